#!/bin/bash
# data-gradient-first ordering in both block executors: parity, then same-box A/B against the committed tree (variants/r03b)
python -m pytest tests/test_block_gpu.py tests/test_half_gpu.py tests/test_step_gpu.py -x -q -m gpu > gpurun_out/r3_t1.log 2>&1 || { tail -40 gpurun_out/r3_t1.log; exit 1; }
tail -2 gpurun_out/r3_t1.log
for i in 1 2 3; do
echo "new fp32:  $(python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-330)"
echo "old fp32:  $(cd variants/r03b && python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-330)"
done
for i in 1 2; do
echo "new half blocks:    $(python bench.py --lean --half --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-330)"
echo "new half per-layer: $(P3D_HALF_BLOCKS=0 python bench.py --lean --half --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-330)"
done

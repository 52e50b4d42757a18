#!/bin/bash
# ReLU mask bytes for the block output: parity + step
python -m pytest tests/test_block_gpu.py tests/test_step_gpu.py -x -q -m gpu > gpurun_out/r3_t1.log 2>&1 || { tail -40 gpurun_out/r3_t1.log; exit 1; }
tail -2 gpurun_out/r3_t1.log
for i in 1 2 3; do
echo "new:  $(python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-330)"
done

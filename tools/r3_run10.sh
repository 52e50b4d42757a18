#!/bin/bash
# K loop with the LDS stores at the head of the step (fetch distance 2): parity of the kernels, per-shape timing and the step against the committed tree (variants/r03b)
python -m pytest tests/test_kernels_gpu.py tests/test_block_gpu.py -x -q -m gpu > gpurun_out/r3_t1.log 2>&1 || { tail -40 gpurun_out/r3_t1.log; exit 1; }
tail -2 gpurun_out/r3_t1.log
python tools/conv_bench.py --img > gpurun_out/r3_cb_new.txt 2>&1
(cd variants/r03b && python tools/conv_bench.py --img) > gpurun_out/r3_cb_old.txt 2>&1
tail -2 gpurun_out/r3_cb_new.txt; tail -2 gpurun_out/r3_cb_old.txt
for i in 1 2 3; do
echo "new fp32:  $(python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-330)"
echo "old fp32:  $(cd variants/r03b && python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-330)"
done

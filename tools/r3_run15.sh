#!/bin/bash
# XCD-aware block order in the weight-gradient kernel: parity, per-shape timing, the step
python -m pytest tests/test_kernels_gpu.py tests/test_block_gpu.py -x -q -m gpu > gpurun_out/r3_t1.log 2>&1 || { tail -40 gpurun_out/r3_t1.log; exit 1; }
tail -2 gpurun_out/r3_t1.log
python tools/conv_bench.py --img --mode wgrad > gpurun_out/r3_cb_new.txt 2>&1
tail -2 gpurun_out/r3_cb_new.txt
for i in 1 2 3; do
echo "fp32:  $(python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-330)"
done

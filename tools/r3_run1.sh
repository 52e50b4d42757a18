#!/bin/bash
# round 3, first GPU call: new image-fed kernels -- parity first, then per-shape timing, then the step against the round-2 tree on the same box
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "image" > gpurun_out/r3_t1.log 2>&1 || { tail -30 gpurun_out/r3_t1.log; exit 1; }
tail -3 gpurun_out/r3_t1.log
python -m pytest tests/test_block_gpu.py tests/test_step_gpu.py -x -q -m gpu > gpurun_out/r3_t2.log 2>&1 || { tail -40 gpurun_out/r3_t2.log; exit 1; }
tail -3 gpurun_out/r3_t2.log
python tools/conv_bench.py --img --iters 20 > gpurun_out/r3_conv_bench1.txt 2>&1 || { tail -20 gpurun_out/r3_conv_bench1.txt; exit 1; }
tail -30 gpurun_out/r3_conv_bench1.txt
echo "new:  $(python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1)" | tee gpurun_out/r3_ab1.txt
echo "r02:  $(python variants/r02/bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1)" | tee -a gpurun_out/r3_ab1.txt
echo "new:  $(python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1)" | tee -a gpurun_out/r3_ab1.txt

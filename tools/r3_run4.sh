#!/bin/bash
# round 3, GPU call 4: LDS-transposed epilogue -- parity, per-shape timing against the previous commit's tree (variants/r03a), step A/B
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_block_gpu.py -x -q -m gpu -k "image or stem or multi_tap or sampled_oracle or x3 or block or trilinear or full_size or partial or masked" > gpurun_out/r3_t1.log 2>&1 || { tail -40 gpurun_out/r3_t1.log; exit 1; }
tail -2 gpurun_out/r3_t1.log
python tools/conv_bench.py --img --iters 20 > gpurun_out/r3_conv_bench4.txt 2>&1 || { tail gpurun_out/r3_conv_bench4.txt; exit 1; }
python variants/r03a/tools/conv_bench.py --img --iters 20 > gpurun_out/r3_conv_bench4_prev.txt 2>&1
echo "--- new"; grep -v amdgpu gpurun_out/r3_conv_bench4.txt | cut -c1-100
echo "--- previous commit"; grep -v amdgpu gpurun_out/r3_conv_bench4_prev.txt | cut -c1-100 | tail -4
echo "new:   $(python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1)" | tee gpurun_out/r3_ab4.txt
echo "prev:  $(python variants/r03a/bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1)" | tee -a gpurun_out/r3_ab4.txt
echo "new:   $(python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1)" | tee -a gpurun_out/r3_ab4.txt

#!/bin/bash
# round 3, GPU call 3: persistent-tile forward / dgrad kernel -- parity, per-shape timing with and without persistence, step A/B
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_block_gpu.py -x -q -m gpu -k "image or stem or multi_tap or sampled_oracle or x3 or block or trilinear or full_size" > gpurun_out/r3_t1.log 2>&1 || { tail -40 gpurun_out/r3_t1.log; exit 1; }
tail -2 gpurun_out/r3_t1.log
python tools/conv_bench.py --img --iters 20 --mode fwd > gpurun_out/r3_conv_bench3.txt 2>&1 || { tail gpurun_out/r3_conv_bench3.txt; exit 1; }
P3D_FX_SLOTS=0 python tools/conv_bench.py --img --iters 20 --mode fwd > gpurun_out/r3_conv_bench3_noper.txt 2>&1
paste -d'\n' <(grep -v amdgpu gpurun_out/r3_conv_bench3.txt | cut -c1-62) /dev/null | grep -v "^$" | head -60
echo "--- one block per tile (P3D_FX_SLOTS=0)"; grep -v amdgpu gpurun_out/r3_conv_bench3_noper.txt | cut -c1-62 | head -60
echo "new:        $(python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1)" | tee gpurun_out/r3_ab3.txt
echo "slots=0:    $(P3D_FX_SLOTS=0 python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1)" | tee -a gpurun_out/r3_ab3.txt
echo "slots=512:  $(P3D_FX_SLOTS=512 python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1)" | tee -a gpurun_out/r3_ab3.txt
echo "r02:        $(python variants/r02/bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1)" | tee -a gpurun_out/r3_ab3.txt

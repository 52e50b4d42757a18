#!/bin/bash
# round 3, GPU call 2: stem + regressor on image operands -- parity, per-shape timing, step A/B against the round-2 tree, then the whole GPU suite
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "stem or multi_tap or image" > gpurun_out/r3_t1.log 2>&1 || { tail -40 gpurun_out/r3_t1.log; exit 1; }
tail -2 gpurun_out/r3_t1.log
python tools/conv_bench.py --img --iters 20 --only "c3 h256" > gpurun_out/r3_conv_bench2.txt 2>&1; python tools/conv_bench.py --img --iters 20 --only "c2048 h16 k272" >> gpurun_out/r3_conv_bench2.txt 2>&1
grep -v amdgpu gpurun_out/r3_conv_bench2.txt
echo "new:  $(python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1)" | tee gpurun_out/r3_ab2.txt
echo "r02:  $(python variants/r02/bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1)" | tee -a gpurun_out/r3_ab2.txt
echo "new, no image convs outside blocks:  $(P3D_IMAGE_CONVS=0 python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1)" | tee -a gpurun_out/r3_ab2.txt
python -m pytest tests -x -q -m gpu > gpurun_out/r3_t_all.log 2>&1 || { tail -60 gpurun_out/r3_t_all.log; exit 1; }
tail -3 gpurun_out/r3_t_all.log

#!/bin/bash
# the whole GPU suite on the current tree, then the per-shape table of profiles/r03_summary.md section 5, then the step with and without the fused stem tail
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k stem_tail > gpurun_out/r3_t0.log 2>&1 || { tail -30 gpurun_out/r3_t0.log; exit 1; }; python -m pytest tests -x -q -m gpu > gpurun_out/r3_t_all.log 2>&1 || { tail -40 gpurun_out/r3_t_all.log; exit 1; }
tail -2 gpurun_out/r3_t_all.log
python tools/conv_bench.py --img --iters 20 > gpurun_out/r3_conv_bench_final.txt 2>&1
tail -2 gpurun_out/r3_conv_bench_final.txt
for i in 1 2 3; do
echo "stem tail fused:  $(python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-330)"
echo "three nodes:      $(P3D_STEM_TAIL=0 python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-330)"
done
echo "r18 fused:  $(python bench.py --lean --model resnet18 --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-330)"
echo "r18 three:  $(P3D_STEM_TAIL=0 python bench.py --lean --model resnet18 --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-330)"

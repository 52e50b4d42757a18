#!/bin/bash
# block order A/B (one change against the class that carries the excess traffic): conv_bench on the layers of layer3 / layer4 / regressor
O=gpurun_out/r4g; mkdir -p $O
for t in "7=0,8=0" "7=1,8=1" "7=0,8=0" "7=1,8=1"; do
  echo "=== tune $t" | tee -a $O/order.txt
  timeout -k 10 300 python tools/conv_bench.py --img --iters 20 --only h16 --tune "$t" 2>&1 | grep -v amdgpu | grep -A1 "^c" | grep -v "^--" >> $O/order.txt
done
timeout -k 10 300 python tools/conv_bench.py --img --iters 20 --tune "7=1,8=1" 2>&1 | grep -v amdgpu > $O/all_order1.txt
timeout -k 10 300 python tools/conv_bench.py --img --iters 20 --tune "7=0,8=0" 2>&1 | grep -v amdgpu > $O/all_order0.txt
tail -2 $O/all_order1.txt; tail -2 $O/all_order0.txt
b() { timeout -k 10 200 python bench.py --lean --steps 30 --warmup 8 "$@" 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
for rep in 1 2 3; do
  echo "built-in order         $(b)" | tee -a $O/ab.txt
done
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_block_gpu.py -x -q -m gpu 2>&1 | tail -3

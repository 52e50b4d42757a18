#!/bin/bash
# round 4: three taps per column tile for the 64 -> 64 3x3 weight gradients (fx_wgrad_kernel<.., TAPS 3>): parity cases, per-shape A/B, step A/B
O=gpurun_out/r4k; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "image_fed or deep_split" > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
grep -q "pytest exit 0" $O/pytest.txt || exit 1
for tg in 256 384 512 640 768 1024; do
  echo "=== three taps, wgrad target $tg: $(timeout -k 10 200 python tools/conv_bench.py --img --iters 30 --only 'c64 h64 k64 3x3' --mode wgrad --tune "9=1,2=$tg" 2>&1 | grep image-fed | sed 's/.*| *\([0-9.]*\) *\([0-9.]*\)   wgrad.*/wgrad \1 ms \2 TF/')" | tee -a $O/three_taps.txt
done
echo "=== two taps (9=2): $(timeout -k 10 200 python tools/conv_bench.py --img --iters 30 --only 'c64 h64 k64 3x3' --mode wgrad --tune "9=2" 2>&1 | grep image-fed | sed 's/.*| *\([0-9.]*\) *\([0-9.]*\)   wgrad.*/wgrad \1 ms \2 TF/')" | tee -a $O/three_taps.txt
for rep in 1 2; do
  for tt in 1 2; do
    echo "P3D_TWO_TAPS=$tt: $(P3D_TWO_TAPS=$tt timeout -k 10 300 python bench.py --lean --steps 30 --warmup 5 2>/dev/null | tail -1 | cut -c1-120)" | tee -a $O/step_ab.txt
  done
done

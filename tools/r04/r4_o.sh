#!/bin/bash
# round 4: the fp32 step against the weight gradients' slab count (percent of the per-shape plan, which was tuned on the kernels alone); and the -half_acc count once more
O=gpurun_out/r4o; mkdir -p $O
b() { timeout -k 10 300 python bench.py --lean --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
for rep in 1 2; do
  for t in 100 75 50 35 150; do
    echo "P3D_WGRAD_SCALE=$t : $(P3D_WGRAD_SCALE=$t b)" | tee -a $O/ab.txt
  done
done
for t in 448 384 320; do
  echo "half P3D_HWGRAD_BLOCKS=$t : $(P3D_HWGRAD_BLOCKS=$t b --half)" | tee -a $O/ab.txt
done

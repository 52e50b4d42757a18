#!/bin/bash
# round 4: split-K block target of forward / data gradient and the weight gradients' slab scale, measured IN the two-stream step (fusionnet batch 32, contract batch 64)
O=gpurun_out/r4r; mkdir -p $O
b() { timeout -k 10 300 python bench.py --lean --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
for t in 768 512 384 0 1024; do
  echo "fusionnet P3D_SPLITK_BLOCKS=$t : $(P3D_SPLITK_BLOCKS=$t b --family fusionnet --batch 32)" | tee -a $O/ab.txt
done
for t in 100 75 50 150; do
  echo "fusionnet P3D_WGRAD_SCALE=$t : $(P3D_WGRAD_SCALE=$t b --family fusionnet --batch 32)" | tee -a $O/ab.txt
done
for t in 768 512 384 0; do
  echo "depthnet P3D_SPLITK_BLOCKS=$t : $(P3D_SPLITK_BLOCKS=$t b)" | tee -a $O/ab.txt
done

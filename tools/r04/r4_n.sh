#!/bin/bash
# round 4: -half_acc weight-gradient slab count (blocks aimed at per layer) in the two-stream step
O=gpurun_out/r4n; mkdir -p $O
b() { timeout -k 10 300 python bench.py --lean --half --steps 30 --warmup 5 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
for rep in 1 2; do
  for t in 1024 768 512 384 256 1536; do
    echo "P3D_HWGRAD_BLOCKS=$t : $(P3D_HWGRAD_BLOCKS=$t b)" | tee -a $O/ab.txt
  done
done

#!/bin/bash
O=gpurun_out/r4s; mkdir -p $O
b() { timeout -k 10 300 python bench.py --lean --half --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
for rep in 1 2 3; do
  echo "mask bytes       : $(b)" | tee -a $O/ab.txt
  echo "P3D_HALF_MASK=0  : $(P3D_HALF_MASK=0 b)" | tee -a $O/ab.txt
done

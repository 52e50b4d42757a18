#!/bin/bash
# round 4: tap-fastest weight-gradient order on the wide multi-tap layers (built-in) against channel-tile-fastest everywhere (P3D_WGRAD_ORDER=0) in the step
O=gpurun_out/r4x2; mkdir -p $O
b() { timeout -k 10 300 python bench.py --lean --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
for rep in 1 2 3; do
  echo "built-in (tap fastest for C >= 512, multi-tap) : $(b)" | tee -a $O/ab.txt
  echo "P3D_WGRAD_ORDER=0                              : $(P3D_WGRAD_ORDER=0 b)" | tee -a $O/ab.txt
done

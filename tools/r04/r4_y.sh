#!/bin/bash
# round 4: traffic-saving orders measured IN the two-stream step: weight-gradient order for all multi-tap layers, tap-inner K order from 512 / 256 reduction channels
O=gpurun_out/r4y2; mkdir -p $O
b() { timeout -k 10 300 python bench.py --lean --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
for rep in 1 2; do
  echo "built-in                      : $(b)" | tee -a $O/ab.txt
  echo "P3D_WGRAD_ORDER=1 (all)       : $(P3D_WGRAD_ORDER=1 b)" | tee -a $O/ab.txt
  echo "P3D_TAP_INNER_MIN=512         : $(P3D_TAP_INNER_MIN=512 b)" | tee -a $O/ab.txt
  echo "P3D_TAP_INNER_MIN=256         : $(P3D_TAP_INNER_MIN=256 b)" | tee -a $O/ab.txt
  echo "P3D_TAP_INNER_MIN=0 (never)   : $(P3D_TAP_INNER_MIN=0 b)" | tee -a $O/ab.txt
done

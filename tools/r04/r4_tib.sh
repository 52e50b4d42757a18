#!/bin/bash
# round 4: tap-inner K order for the DATA gradients from 512 reduction channels (layer4's 3x3 layers) in the two-stream step -- their fetches compete with the weight gradients' there
b() { timeout -k 10 300 python bench.py --lean --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
for rep in 1 2 3 4 5 6; do
  echo "default (1024)              : $(b)"
  echo "P3D_TAP_INNER_MIN_BWD=512   : $(P3D_TAP_INNER_MIN_BWD=512 b)"
done

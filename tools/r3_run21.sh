#!/bin/bash
# how the weight-gradient stream is created: probe (default) vs low / high priority
for i in 1 2; do
for m in probe low high; do
echo "$m:  $(P3D_SIDE_STREAM=$m python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-330)"
done; done

#!/bin/bash
# round 4: -half_acc weight gradient, column tiles tap-fastest on the wide multi-tap layers: tests, step A/B
O=gpurun_out/r4z; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_half_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -2 $O/pytest.txt
grep -q "pytest exit 0" $O/pytest.txt || exit 1
b() { timeout -k 10 300 python bench.py --lean --half --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
for rep in 1 2 3; do
  echo "built-in (C >= 512)       : $(b)" | tee -a $O/ab.txt
  echo "P3D_HWGRAD_TAP_FAST=0     : $(P3D_HWGRAD_TAP_FAST=0 b)" | tee -a $O/ab.txt
  echo "P3D_HWGRAD_TAP_FAST=1 all : $(P3D_HWGRAD_TAP_FAST=1 b)" | tee -a $O/ab.txt
done

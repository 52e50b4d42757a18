#!/bin/bash
# Sweep the six tile shapes over the ResNet-50 layer classes (one process per shape: the override is read once).
for cfg in 0 1 2 3 4 5; do
  echo "=== P3D_FORCE_CFG=$cfg"
  P3D_FORCE_CFG=$cfg timeout -k 10 300 python tools/conv_bench.py "$@" 2>&1 | grep -v amdgpu.ids
done

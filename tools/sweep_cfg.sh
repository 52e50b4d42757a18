#!/bin/bash
# Sweep tile shapes over the ResNet-50 layer classes (one process per shape: the override is read once).
# usage: tools/sweep_cfg.sh "0 6 7" [conv_bench args]
cfgs="$1"; shift
echo "=== default"; timeout -k 10 300 python tools/conv_bench.py "$@" 2>&1 | grep -v amdgpu.ids
for cfg in $cfgs; do
  echo "=== P3D_FORCE_CFG=$cfg"
  P3D_FORCE_CFG=$cfg timeout -k 10 300 python tools/conv_bench.py "$@" 2>&1 | grep -v amdgpu.ids
done

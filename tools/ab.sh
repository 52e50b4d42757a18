#!/bin/bash
# A/B tuning builds in ONE box: tools/ab.sh "<conv_bench args>" lib1.so lib2.so ...  (first pass = shipped library)
args="$1"; shift
echo "=== shipped"; timeout -k 10 300 python tools/conv_bench.py $args 2>&1 | grep -v amdgpu.ids
for lib in "$@"; do echo "=== $lib"; P3D_LIB=$PWD/$lib timeout -k 10 300 python tools/conv_bench.py $args 2>&1 | grep -v amdgpu.ids; done
echo "=== shipped (again)"; timeout -k 10 300 python tools/conv_bench.py $args 2>&1 | grep -v amdgpu.ids

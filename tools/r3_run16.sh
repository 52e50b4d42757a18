#!/bin/bash
# new slab plan: wgrad per-shape + step; then the fp16 step's kernel table (serialised)
python tools/conv_bench.py --img --mode wgrad 2>&1 | tail -1
for i in 1 2; do
echo "fp32:  $(python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-330)"
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_half
P3D_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_half -o p -- python3 bench.py --steps 5 --warmup 2 --lean --half > gpurun_out/prof_half.log 2>&1
python3 tools/kernel_table.py gpurun_out/prof_half 7 | head -40

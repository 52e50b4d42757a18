#!/bin/bash
O=gpurun_out/r4f; mkdir -p $O
timeout -k 10 300 python tools/conv_bench.py --img --iters 20 --batch 32 > $O/conv_b32.txt 2>&1; tail -4 $O/conv_b32.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P3D_WGRAD_STREAM=0 P3D_BLOCK_SIDE=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fusion -o p -- python3 bench.py --steps 5 --warmup 2 --lean --family fusionnet --batch 32 > $O/prof_fusion.log 2>&1
python3 tools/kernel_table.py $O/prof_fusion 7 > $O/fusion_serial_table.txt; head -45 $O/fusion_serial_table.txt
P3D_WGRAD_STREAM=0 P3D_BLOCK_SIDE=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_partial -o p -- python3 bench.py --steps 5 --warmup 2 --lean --family partial_depthnet > $O/prof_partial.log 2>&1
python3 tools/kernel_table.py $O/prof_partial 7 > $O/partial_serial_table.txt; head -40 $O/partial_serial_table.txt
rm -rf $O/prof_fusion $O/prof_partial

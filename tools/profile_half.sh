#!/bin/bash
# rocprofv3 kernel stats of the -half_acc step (bench.py --half --lean, 7 identical steps), every kernel on one stream and with the product's two streams
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; rm -rf gpurun_out/prof_$name; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -o p -- python3 bench.py --steps 5 --warmup 2 --lean --half > gpurun_out/prof_$name.log 2>&1; }
P3D_WGRAD_STREAM=0 P3D_BLOCK_SIDE=0 run half_serial || exit 1
run half_overlap || exit 1
python3 tools/kernel_table.py gpurun_out/prof_half_serial 7 > gpurun_out/half_serial_table.txt; cat gpurun_out/half_serial_table.txt
python3 tools/kernel_table.py gpurun_out/prof_half_overlap 7 > gpurun_out/half_overlap_table.txt; tail -1 gpurun_out/half_overlap_table.txt
tail -1 gpurun_out/prof_half_serial.log | cut -c1-300; tail -1 gpurun_out/prof_half_overlap.log | cut -c1-300

#!/bin/bash
# Round profile set (run on the GPU box through gpurun): rocprofv3 kernel stats of bench.py in three configurations + the PMC traffic pass.
#   serial  : P3D_WGRAD_STREAM=0  -- per-launch durations are each kernel alone (what roofline.achieved is computed from)
#   overlap : product default      -- wgrad kernels on the second stream
#   half    : --half               -- fp16 NHWC path
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; rm -rf gpurun_out/prof_$name; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/prof_$name.log 2>&1; }
P3D_WGRAD_STREAM=0 run serial && run overlap && run half --half && bash tools/traffic.sh > gpurun_out/traffic.log 2>&1
ls gpurun_out/prof_serial/* | head

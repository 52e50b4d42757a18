#!/bin/bash
# Round-4 profile set (run on the GPU box through gpurun): rocprofv3 kernel stats of bench.py --lean (7 identical steps: 2 warm-up + 5 timed).
#   serial  : P3D_WGRAD_STREAM=0 P3D_BLOCK_SIDE=0 -- every kernel alone on the GPU (per-launch durations = what roofline.achieved is computed from)
#   overlap : product default -- weight-gradient kernels on the second stream        (only with "all")
#   pmc     : FETCH_SIZE / WRITE_SIZE passes over the serialised command, per dispatch (only with "all"): split by kernel class and layer group by summarize_r04.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; rm -rf gpurun_out/prof_$name; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -o p -- python3 bench.py --steps 5 --warmup 2 --lean > gpurun_out/prof_$name.log 2>&1; }
P3D_WGRAD_STREAM=0 P3D_BLOCK_SIDE=0 run serial || exit 1
if [ "$1" = "all" ]; then
  run overlap || exit 1
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_$c
    P3D_WGRAD_STREAM=0 P3D_BLOCK_SIDE=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 2 --warmup 1 --lean > gpurun_out/pmc_$c.log 2>&1 || exit 1
  done
fi
python3 tools/kernel_table.py gpurun_out/prof_serial 7 > gpurun_out/r4_serial_table.txt; cat gpurun_out/r4_serial_table.txt; [ "$1" = "all" ] && python3 tools/summarize_r04.py

#!/bin/bash
# rocprofv3 kernel stats of one family's lean step: tools/profile_family.sh <name> <bench.py arguments...>; serial (one stream) table to gpurun_out/<name>_serial_table.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=$1; shift
rm -rf gpurun_out/prof_$name
P3D_WGRAD_STREAM=0 P3D_BLOCK_SIDE=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -o p -- python3 bench.py --steps 5 --warmup 2 --lean "$@" > gpurun_out/prof_$name.log 2>&1 || exit 1
python3 tools/kernel_table.py gpurun_out/prof_$name 7 > gpurun_out/${name}_serial_table.txt; cat gpurun_out/${name}_serial_table.txt

#!/bin/bash
python -m pytest tests/test_half_gpu.py -x -q -m gpu > gpurun_out/r3_t1.log 2>&1 || { tail -40 gpurun_out/r3_t1.log; exit 1; }
tail -2 gpurun_out/r3_t1.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_half
P3D_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_half -o p -- python3 bench.py --steps 5 --warmup 2 --lean --half > gpurun_out/prof_half.log 2>&1
python3 tools/kernel_table.py gpurun_out/prof_half 7 | grep -i "finalize\|all kernels"
for i in 1 2; do echo "half: $(python bench.py --lean --half --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-200)"; done

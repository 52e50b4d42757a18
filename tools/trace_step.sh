# kernel trace of the product configuration (two streams): bench.py --lean, 5 timed steps after 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace_step
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_step -o t -- python3 bench.py --steps 5 --warmup 3 --lean > gpurun_out/trace_step.log 2>&1
tail -1 gpurun_out/trace_step.log | cut -c1-200

#!/bin/bash
# round 4, call C: tests of the changed paths, strided-dgrad one-launch A/B, wgrad-late A/B, a kernel trace of the two-stream step with its timeline
set -o pipefail
O=gpurun_out/r4c; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_block_gpu.py "tests/test_step_gpu.py::test_graph_recapture_after_a_learning_rate_change" "tests/test_step_gpu.py::test_whole_step_graph_capture_matches_eager" "tests/test_step_gpu.py::test_training_is_bitwise_reproducible" -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -4 $O/pytest.txt
b() { timeout -k 10 200 python bench.py --lean --steps 30 --warmup 8 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
for rep in 1 2 3; do
  echo "default                $(b)" | tee -a $O/ab.txt
  echo "P3D_WGRAD_LATE=1       $(P3D_WGRAD_LATE=1 b)" | tee -a $O/ab.txt
  echo "P3D_FX16=0             $(P3D_FX16=0 b)" | tee -a $O/ab.txt
done
timeout -k 10 200 python tools/conv_bench.py --img --iters 20 --only s2 > $O/conv_s2.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 bench.py --steps 5 --warmup 3 --lean > $O/trace.log 2>&1
f=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python tools/timeline.py $f 2 > $O/timeline.txt 2>&1; head -60 $O/timeline.txt
rm -rf $O/trace

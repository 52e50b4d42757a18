#!/bin/bash
set -o pipefail
O=gpurun_out/r4b; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_rccl_gpu.py "tests/test_kernels_gpu.py::test_extra_channel_stem_forward_backward" "tests/test_kernels_gpu.py::test_stem_on_the_x3_kernels" "tests/test_step_gpu.py::test_contract_batch_step_matches_oracle" -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -15 $O/pytest.txt
bash tools/cumask_sweep.sh

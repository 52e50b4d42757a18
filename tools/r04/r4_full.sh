#!/bin/bash
set -o pipefail
O=gpurun_out/r4full; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
grep -E "passed|failed|FAILED|ERROR" $O/pytest.txt | tail -20

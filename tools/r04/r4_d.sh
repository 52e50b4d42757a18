#!/bin/bash
set -o pipefail
O=gpurun_out/r4d; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_block_gpu.py tests/test_step_gpu.py tests/test_kernels_gpu.py::test_large_filter_weight_gradient_with_deep_split -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -4 $O/pytest.txt
b() { timeout -k 10 200 python bench.py --lean --steps 30 --warmup 8 "$@" 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
for rep in 1 2 3; do
  echo "default                $(b)" | tee -a $O/ab.txt
  echo "P3D_TAIL_SUMS=0        $(P3D_TAIL_SUMS=0 b)" | tee -a $O/ab.txt
done
echo "r18 default            $(b --model resnet18)" | tee -a $O/ab.txt
echo "r18 P3D_TAIL_SUMS=0    $(P3D_TAIL_SUMS=0 b --model resnet18)" | tee -a $O/ab.txt

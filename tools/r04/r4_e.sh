#!/bin/bash
set -o pipefail
O=gpurun_out/r4e; mkdir -p $O
timeout -k 10 900 python -m pytest "tests/test_block_gpu.py::test_masked_block_on_the_executor_matches_the_per_layer_path" "tests/test_kernels_gpu.py::test_partial_conv_stem_on_the_restated_kernels" "tests/test_kernels_gpu.py::test_partial_conv_matches_reference_golden" "tests/test_kernels_gpu.py::test_conv_cat_equals_conv_of_concat" "tests/test_kernels_gpu.py::test_conv_cat_sampled_oracle_at_fusion_size" tests/test_step_gpu.py -q -m gpu > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -30 $O/pytest.txt | grep -v Warning | tail -12
b() { timeout -k 10 200 python bench.py --lean --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
for rep in 1 2; do
  echo "partial_fusionnet r50 bs32                     $(b --family partial_fusionnet --batch 32)" | tee -a $O/ab2.txt
  echo "partial_fusionnet r50 bs32 BLOCKS=0 STEM=0     $(P3D_MASKED_BLOCKS=0 P3D_MASKED_STEM=0 b --family partial_fusionnet --batch 32)" | tee -a $O/ab2.txt
  echo "fusionnet r50 bs32                             $(b --family fusionnet --batch 32)" | tee -a $O/ab2.txt
done

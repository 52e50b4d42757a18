#!/bin/bash
# round 4, call A: fx16 (16x16x32 MFMA) correctness + A/B against the 32x32x16 kernels in one box
set -o pipefail
O=gpurun_out/r4a; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py tests/test_block_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
tail -5 $O/pytest.txt
timeout -k 10 300 python tools/conv_bench.py --img --iters 20 > $O/conv_fx16.txt 2>&1 && echo fx16 done
P3D_FX16=0 timeout -k 10 300 python tools/conv_bench.py --img --iters 20 > $O/conv_base.txt 2>&1 && echo base done
for i in 1 2; do
  timeout -k 10 200 python bench.py --lean --steps 30 --warmup 8 2>/dev/null | tail -1 | cut -c1-200 | sed 's/^/fx16: /' | tee -a $O/lean.txt
  P3D_FX16=0 timeout -k 10 200 python bench.py --lean --steps 30 --warmup 8 2>/dev/null | tail -1 | cut -c1-200 | sed 's/^/base: /' | tee -a $O/lean.txt
done

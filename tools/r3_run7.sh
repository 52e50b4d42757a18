#!/bin/bash
# fused finalizes + batched weight images: parity (blocks, steps, reproducibility, distill / eval users of the executor), step A/B against the previous commit
python -m pytest tests/test_block_gpu.py tests/test_step_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "block or step or reproducible or contract or image or stem or multi_tap" > gpurun_out/r3_t1.log 2>&1 || { tail -40 gpurun_out/r3_t1.log; exit 1; }
tail -2 gpurun_out/r3_t1.log
for i in 1 2; do
echo "new:  $(python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-400)"
echo "prev: $(python variants/r03a/bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-400)"
done
python -m pytest tests -x -q -m gpu -k "not block_gpu and not test_step_gpu and not test_kernels_gpu" > gpurun_out/r3_t2.log 2>&1 || { tail -40 gpurun_out/r3_t2.log; exit 1; }
tail -2 gpurun_out/r3_t2.log

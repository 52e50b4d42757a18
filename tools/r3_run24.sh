#!/bin/bash
# one-launch slab reduction for deep splits: parity, per-shape wgrad, contract bench numbers for the weight-gradient leg
python -m pytest tests/test_kernels_gpu.py tests/test_block_gpu.py tests/test_step_gpu.py -x -q -m gpu > gpurun_out/r3_t1.log 2>&1 || { tail -40 gpurun_out/r3_t1.log; exit 1; }
tail -2 gpurun_out/r3_t1.log
python tools/conv_bench.py --img --mode wgrad 2>&1 | tail -1
for i in 1 2; do
echo "new:  $(python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-330)"
done
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"

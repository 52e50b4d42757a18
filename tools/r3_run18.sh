#!/bin/bash
python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -x -q -m gpu > gpurun_out/r3_t1.log 2>&1 || { tail -40 gpurun_out/r3_t1.log; exit 1; }
tail -2 gpurun_out/r3_t1.log
python tools/conv_bench.py --img --only "c3 h256" 2>&1 | grep -v amdgpu.ids | head -4
for i in 1 2 3; do
echo "new:  $(python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-330)"
done

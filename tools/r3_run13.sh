#!/bin/bash
# plan-owned buffers in the block executor: parity, then repeated lean runs (is the multi-second hipMalloc step gone?), LDS-DMA on / off
python -m pytest tests/test_block_gpu.py tests/test_step_gpu.py -x -q -m gpu > gpurun_out/r3_t1.log 2>&1 || { tail -40 gpurun_out/r3_t1.log; exit 1; }
tail -2 gpurun_out/r3_t1.log
for i in 1 2 3 4; do
echo "dma:  $(P3D_BENCH_WATCHDOG=1 python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c165-700)"
echo "reg:  $(P3D_FX_DMA=0 P3D_BENCH_WATCHDOG=1 python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c165-700)"
done

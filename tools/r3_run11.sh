#!/bin/bash
# LDS-DMA instances (AMODE 2) of the forward / data-gradient kernel: parity, per-shape timing and the step with and without them (P3D_FX_DMA=0) in one binary
python -m pytest tests/test_kernels_gpu.py tests/test_block_gpu.py -x -q -m gpu > gpurun_out/r3_t1.log 2>&1 || { tail -40 gpurun_out/r3_t1.log; exit 1; }
tail -2 gpurun_out/r3_t1.log
python tools/conv_bench.py --img > gpurun_out/r3_cb_dma.txt 2>&1
P3D_FX_DMA=0 python tools/conv_bench.py --img > gpurun_out/r3_cb_reg.txt 2>&1
tail -1 gpurun_out/r3_cb_dma.txt; tail -1 gpurun_out/r3_cb_reg.txt
for i in 1 2 3; do
echo "dma:  $(python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-330)"
echo "reg:  $(P3D_FX_DMA=0 python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-330)"
done

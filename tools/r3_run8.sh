#!/bin/bash
# fp16 block executor: parity, then the fp16 informational line with and without it; RCCL single-rank rehearsal with NCCL_ALGO=Ring
python -m pytest tests/test_half_gpu.py -x -q -m gpu > gpurun_out/r3_t1.log 2>&1 || { tail -40 gpurun_out/r3_t1.log; exit 1; }
tail -2 gpurun_out/r3_t1.log
for i in 1 2; do
echo "half blocks:    $(python bench.py --lean --half --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-330)"
echo "half per-layer: $(P3D_HALF_BLOCKS=0 python bench.py --lean --half --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-330)"
done
echo "rccl 1-rank:    $(P3D_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --lean --steps 10 --warmup 3 2>&1 | tail -1 | cut -c75-330)"
python tools/cpu_bound.py --half 2>&1 | tail -3

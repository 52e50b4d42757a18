#!/bin/bash
# whole GPU suite on the current tree, then lean runs (stall check) and the other BASELINE configurations
python -m pytest tests -x -q -m gpu > gpurun_out/r3_t_all.log 2>&1 || { tail -40 gpurun_out/r3_t_all.log; exit 1; }
tail -2 gpurun_out/r3_t_all.log
for i in 1 2 3 4; do
echo "fp32:  $(python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-700)"
done
echo "half:  $(python bench.py --lean --half --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-700)"
echo "half per-layer:  $(P3D_HALF_BLOCKS=0 python bench.py --lean --half --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-700)"

#!/bin/bash
# wgrad slab plan A/B (same box, interleaved) + the whole GPU suite
for i in 1 2; do
echo "new:  $(python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-150)"
echo "prev: $(python variants/r03a/bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-150)"
done
python -m pytest tests -x -q -m gpu > gpurun_out/r3_t_all.log 2>&1 || { tail -60 gpurun_out/r3_t_all.log; exit 1; }
tail -3 gpurun_out/r3_t_all.log

#!/bin/bash
# stream-priority experiment + step A/B of the transposed epilogue (same box, interleaved)
for i in 1 2; do
echo "new probe:  $(python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-150)"
echo "prev probe: $(python variants/r03a/bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-150)"
echo "new low:    $(P3D_SIDE_STREAM=low python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-150)"
echo "new torch:  $(P3D_SIDE_STREAM=torch python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-150)"
echo "one stream: $(P3D_WGRAD_STREAM=0 P3D_BLOCK_SIDE=0 python bench.py --lean --steps 30 --warmup 8 2>&1 | tail -1 | cut -c75-150)"
done

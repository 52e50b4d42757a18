#!/bin/bash
# where is the multi-second step?  lean runs with a watchdog that dumps the Python stack of a step whose enqueue takes > 0.3 s
for i in 1 2 3; do
P3D_BENCH_WATCHDOG=1 python bench.py --lean --steps 30 --warmup 8 > gpurun_out/r3_wd_$i.log 2>&1
tail -1 gpurun_out/r3_wd_$i.log | cut -c165-700
done
grep -h -A25 "most recent call first" gpurun_out/r3_wd_*.log | head -80

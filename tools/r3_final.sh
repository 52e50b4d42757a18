#!/bin/bash
# round 3 measurement set: profiles (serial / overlap / PMC), the contract bench line, the informational lines of the other configurations
mkdir -p gpurun_out
bash tools/profile_r03.sh all > gpurun_out/r3_profile.log 2>&1 || { tail -20 gpurun_out/r3_profile.log; exit 1; }
tail -45 gpurun_out/r3_profile.log
python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err || { tail -20 gpurun_out/r3_bench.err; exit 1; }
cat gpurun_out/r3_bench.json
bash tools/other_lines.sh > gpurun_out/r3_other_configs.txt 2>&1; cat gpurun_out/r3_other_configs.txt

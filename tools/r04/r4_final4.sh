#!/bin/bash
# round 4 final, part 4: ONE box for the contract line AND its profile set (so that the bench line and the rocprofv3 tables agree): probe the box with a lean run first
O=gpurun_out/r4final; mkdir -p $O
probe=$(python bench.py --lean --steps 20 --warmup 5 2>/dev/null | tail -1 | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1/')
echo "probe: $probe ms per step" | tee $O/probe.txt
if [ "$1" != "any" ] && python -c "import sys; sys.exit(0 if float('$probe') > 27.95 else 1)"; then echo "box at the slow end of the pool: not used for the committed set"; exit 0; fi
timeout -k 10 900 python -m pytest tests/test_bench_contract.py -x -q -m gpu > $O/pytest_bench.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest_bench.txt
grep -q "pytest exit 0" $O/pytest_bench.txt || exit 1
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
bash tools/profile_r04.sh all > $O/profile.log 2>&1; tail -2 $O/profile.log | cut -c1-200
bash tools/other_lines.sh 2>&1 | tee $O/other_configs.txt
b() { python bench.py --lean --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
echo "partial_depthnet r50 bs64 half   $(b --half --family partial_depthnet)" | tee -a $O/other_configs.txt
echo "fusionnet r50 bs32 half          $(b --half --family fusionnet --batch 32)" | tee -a $O/other_configs.txt
bash tools/profile_half.sh > $O/profile_half.log 2>&1; tail -3 $O/profile_half.log | cut -c1-200
echo "complete" | tee -a $O/probe.txt

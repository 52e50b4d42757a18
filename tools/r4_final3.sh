#!/bin/bash
# round 4 final, part 3: the bench contract test, the contract line, other configurations, the profile set (fp32 + half)
O=gpurun_out/r4final; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_bench_contract.py -x -q -m gpu > $O/pytest_bench.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest_bench.txt
tail -2 $O/pytest_bench.txt
grep -q "pytest exit 0" $O/pytest_bench.txt || exit 1
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
bash tools/other_lines.sh 2>&1 | tee $O/other_configs.txt
b() { python bench.py --lean --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
echo "partial_depthnet r50 bs64 half   $(b --half --family partial_depthnet)" | tee -a $O/other_configs.txt
echo "fusionnet r50 bs32 half          $(b --half --family fusionnet --batch 32)" | tee -a $O/other_configs.txt
bash tools/profile_r04.sh all > $O/profile.log 2>&1; tail -3 $O/profile.log | cut -c1-300
bash tools/profile_half.sh > $O/profile_half.log 2>&1; tail -3 $O/profile_half.log | cut -c1-200

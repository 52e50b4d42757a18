#!/bin/bash
# round 4 final, part 1: the full GPU suite exactly as the driver runs it, smoke, the contract bench line
O=gpurun_out/r4final; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
grep -E "passed|failed" $O/pytest.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json

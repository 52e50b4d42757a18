#!/bin/bash
O=gpurun_out/r4j; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_block_gpu.py tests/test_step_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_tail.txt
bash tools/profile_r04.sh > $O/profile_serial.log 2>&1; grep -E "block_close_fwd|block_open_bwd|fx_act_image|all kernels" $O/profile_serial.log
b() { timeout -k 10 200 python bench.py --lean --steps 30 --warmup 8 "$@" 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
for rep in 1 2 3; do echo "default                $(b)" | tee -a $O/ab.txt; done

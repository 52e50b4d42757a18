#!/bin/bash
python -m pytest tests -x -q -m gpu > gpurun_out/r3_t_all.log 2>&1 || { tail -40 gpurun_out/r3_t_all.log; exit 1; }
tail -2 gpurun_out/r3_t_all.log
bash tools/r3_final.sh

#!/bin/bash
# round 4 final, part 2: the other configurations' lines and the profile set
O=gpurun_out/r4final; mkdir -p $O
bash tools/other_lines.sh 2>&1 | tee $O/other_configs.txt
bash tools/profile_r04.sh all > $O/profile.log 2>&1; tail -30 $O/profile.log

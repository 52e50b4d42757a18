#!/bin/bash
# weight-gradient stream (and optionally the launch stream) restricted to a share of the compute units; lean bench, one box, interleaved
O=gpurun_out/r4b; mkdir -p $O
python tools/cumask_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/probe.txt
b() { timeout -k 10 200 python bench.py --lean --steps 30 --warmup 8 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
for rep in 1 2; do
  echo "default (probe stream)        $(b)" | tee -a $O/sweep.txt
  for n in 224 192 160 128 96; do
    echo "side $n                      $(P3D_SIDE_CUS=$n b)" | tee -a $O/sweep.txt
  done
  echo "side 64:hi main 192           $(P3D_SIDE_CUS=64:hi P3D_MAIN_CUS=192 b)" | tee -a $O/sweep.txt
  echo "side 96:hi main 160           $(P3D_SIDE_CUS=96:hi P3D_MAIN_CUS=160 b)" | tee -a $O/sweep.txt
  echo "side 128:hi main 256          $(P3D_SIDE_CUS=128:hi P3D_MAIN_CUS=256 b)" | tee -a $O/sweep.txt
  echo "side 192:hi main 224          $(P3D_SIDE_CUS=192:hi P3D_MAIN_CUS=224 b)" | tee -a $O/sweep.txt
done

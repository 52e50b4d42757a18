#!/bin/bash
# round 4: the first layer's weight gradient (the last kernel of backward) on the launch stream instead of the second stream
O=gpurun_out/r4t; mkdir -p $O
b() { timeout -k 10 300 python bench.py --lean --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
for rep in 1 2 3; do
  echo "depthnet  main stream : $(b)" | tee -a $O/ab.txt
  echo "depthnet  side stream : $(P3D_LAST_WGRAD_MAIN=0 b)" | tee -a $O/ab.txt
done
for rep in 1 2; do
  echo "half      main stream : $(b --half)" | tee -a $O/ab.txt
  echo "half      side stream : $(P3D_LAST_WGRAD_MAIN=0 b --half)" | tee -a $O/ab.txt
  echo "fusionnet main stream : $(b --family fusionnet --batch 32)" | tee -a $O/ab.txt
  echo "fusionnet side stream : $(P3D_LAST_WGRAD_MAIN=0 b --family fusionnet --batch 32)" | tee -a $O/ab.txt
  echo "r18 bs8   main stream : $(b --model resnet18 --batch 8)" | tee -a $O/ab.txt
  echo "r18 bs8   side stream : $(P3D_LAST_WGRAD_MAIN=0 b --model resnet18 --batch 8)" | tee -a $O/ab.txt
done

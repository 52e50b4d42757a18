#!/bin/bash
# round 4: weight images rebuilt on the second stream right behind the optimizer (beside the next step's stem) instead of on the launch stream in front of block 1
O=gpurun_out/r4u; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_step_gpu.py -x -q -m gpu -k "reproducible or contract or golden or oracle" > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/pytest.txt
P3D_IMAGES_EARLY=1 timeout -k 10 600 python -m pytest tests/test_step_gpu.py -x -q -m gpu > $O/pytest_early.txt 2>&1; echo "pytest (early) exit $?" | tee -a $O/pytest_early.txt
tail -2 $O/pytest_early.txt
b() { timeout -k 10 300 python bench.py --lean --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
for rep in 1 2 3; do
  echo "default            : $(b)" | tee -a $O/ab.txt
  echo "P3D_IMAGES_EARLY=1 : $(P3D_IMAGES_EARLY=1 b)" | tee -a $O/ab.txt
done
echo "r18 bs8 default            : $(b --model resnet18 --batch 8)" | tee -a $O/ab.txt
echo "r18 bs8 P3D_IMAGES_EARLY=1 : $(P3D_IMAGES_EARLY=1 b --model resnet18 --batch 8)" | tee -a $O/ab.txt

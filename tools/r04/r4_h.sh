#!/bin/bash
O=gpurun_out/r4h; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_block_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_tail.txt
for t in "9=1" "9=0" "9=1" "9=0"; do
  echo "=== tune $t" | tee -a $O/two_taps.txt
  timeout -k 10 200 python tools/conv_bench.py --img --iters 30 --only "c64 h64 k64 3x3" --tune "$t" 2>&1 | grep -v amdgpu | grep -A1 "^c" >> $O/two_taps.txt
done
cat $O/two_taps.txt | cut -c1-150
b() { timeout -k 10 200 python bench.py --lean --steps 30 --warmup 8 "$@" 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
for rep in 1 2 3; do
  echo "default                $(b)" | tee -a $O/ab.txt
  echo "P3D_TWO_TAPS=0         $(P3D_TWO_TAPS=0 b)" | tee -a $O/ab.txt
done
echo "r18 default            $(b --model resnet18)" | tee -a $O/ab.txt
echo "r18 P3D_TWO_TAPS=0     $(P3D_TWO_TAPS=0 b --model resnet18)" | tee -a $O/ab.txt
echo "r18 default            $(b --model resnet18)" | tee -a $O/ab.txt
echo "r18 P3D_TWO_TAPS=0     $(P3D_TWO_TAPS=0 b --model resnet18)" | tee -a $O/ab.txt
echo "partial_depthnet r50   $(b --family partial_depthnet)" | tee -a $O/ab.txt
# FETCH_SIZE of the regressor's forward launch under the two block orders
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for o in 0 1; do
  rm -rf $O/pmc_o$o
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_o$o -- python3 tools/conv_bench.py --only "k272" --mode fwd --img --iters 3 --tune "7=$o" > $O/pmc_o$o.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob('$O/pmc_o$o/*/*_counter_collection.csv')[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'fx16_conv_kernel' in n or 'fx_conv_kernel' in n:
        agg[n[:60]].append(float(r['Counter_Value']))
for k, v in agg.items():
    print('order $o  %-60s FETCH_SIZE mean %.0f KB over %d launches' % (k, sum(v) / len(v), len(v)))
PY
  rm -rf $O/pmc_o$o
done 2>&1 | tee $O/regressor_fetch.txt

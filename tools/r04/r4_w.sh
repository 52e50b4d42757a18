#!/bin/bash
# round 4: the weight gradient's block order (tap fastest: p3d_fx_tune(8, 1)) on the wide multi-tap layers: FETCH_SIZE per launch and time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4w; mkdir -p $O; : > $O/wgrad_order.txt
for only in "c2048 h16 k272" "c512 h16 k512 3x3 s1 d1"; do
  for t in "8=0" "8=1"; do
    rm -rf $O/pmc
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc -- python3 tools/conv_bench.py --only "$only" --img --iters 3 --tune "$t" > $O/pmc.log 2>&1
    python3 - "$only" "$t" >> $O/wgrad_order.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob('gpurun_out/r4w/pmc/*/*_counter_collection.csv')
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if 'fx_wgrad_kernel<true, true' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE': agg[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
for k, v in agg.items(): print('%-26s tune %s  %s  FETCH_SIZE mean %.0f KB over %d launches' % (sys.argv[1], sys.argv[2], k, sum(v) / len(v), len(v)))
PY
    echo "$only tune $t : $(timeout -k 10 200 python tools/conv_bench.py --img --iters 30 --only "$only" --tune "$t" 2>&1 | grep image-fed | sed 's/.*| *\([0-9.]*\) *\([0-9.]*\)   wgrad.*/wgrad \1 ms \2 TF/')" >> $O/wgrad_order.txt
  done
done
cat $O/wgrad_order.txt

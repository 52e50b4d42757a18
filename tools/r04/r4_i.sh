#!/bin/bash
O=gpurun_out/r4i; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_block_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_tail.txt
for t in "10=0" "10=512" "10=256" "10=0" "10=512" "10=256"; do
  echo "=== tune $t" >> $O/tapi.txt
  timeout -k 10 300 python tools/conv_bench.py --img --iters 20 --only "3x3" --tune "$t" 2>&1 | grep -v amdgpu | grep -A1 "^c" | grep -v "^--" >> $O/tapi.txt
done
python3 - <<'PY'
import re, collections
cur=None; rows=collections.defaultdict(lambda: collections.defaultdict(list)); shape=None
for l in open('gpurun_out/r4i/tapi.txt'):
    if l.startswith('==='): cur=l.split()[2]; continue
    m=re.match(r'^(c\d+ h\d+ k\d+ \dx\d s\d d\d)',l)
    if m: shape=m.group(1); continue
    if l.strip().startswith('image-fed'):
        nums=[float(a) for a,b in re.findall(r'\|\s+([\d.]+)\s+([\d.]+)',l)]
        rows[shape][cur].append(nums[:2])
for s,d in rows.items():
    print('%-26s' % s, '  '.join('%s: fwd %s dgrad %s' % (k, '/'.join('%.3f'%v[0] for v in d[k]), '/'.join('%.3f'%v[1] for v in d[k])) for k in d))
PY
b() { timeout -k 10 200 python bench.py --lean --steps 30 --warmup 8 "$@" 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
for rep in 1 2 3; do echo "default                $(b)" | tee -a $O/ab.txt; done
for tg in 256 384 512 768 1024 1536; do
  echo "=== two taps, wgrad target $tg: $(timeout -k 10 200 python tools/conv_bench.py --img --iters 30 --only 'c64 h64 k64 3x3' --mode wgrad --tune "9=1,2=$tg" 2>&1 | grep image-fed | sed 's/.*| *\([0-9.]*\) *\([0-9.]*\)   wgrad.*/wgrad \1 ms \2 TF/')" | tee -a $O/two_taps_targets.txt
done
echo "=== one tap (9=0): $(timeout -k 10 200 python tools/conv_bench.py --img --iters 30 --only 'c64 h64 k64 3x3' --mode wgrad --tune "9=0" 2>&1 | grep image-fed | sed 's/.*| *\([0-9.]*\) *\([0-9.]*\)   wgrad.*/wgrad \1 ms \2 TF/')" | tee -a $O/two_taps_targets.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for o in 0 512; do
  for sh in "k272" "c512 h16 k512 3x3 s1 d1"; do
    rm -rf $O/pmc_t
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_t -- python3 tools/conv_bench.py --only "$sh" --mode fwd --img --iters 3 --tune "10=$o" > $O/pmc_t.log 2>&1
    python3 - <<PY
import csv, glob, collections
f = glob.glob('$O/pmc_t/*/*_counter_collection.csv')[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'fx16_conv_kernel' in n or 'fx_conv_kernel<1' in n:
        agg[n[:70]].append(float(r['Counter_Value']))
for k, v in agg.items():
    print('tap-inner min $o  $sh  %-70s FETCH_SIZE mean %.0f KB over %d launches' % (k, sum(v) / len(v), len(v)))
PY
  done
done 2>&1 | tee $O/tapi_fetch.txt
rm -rf $O/pmc_t

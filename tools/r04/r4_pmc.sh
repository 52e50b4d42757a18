#!/bin/bash
# round 4: clock, matrix-pipe occupancy and LDS bank conflicts of the new kernel instances (one rocprofv3 --pmc pass per shape over tools/conv_bench.py --img)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4pmc; mkdir -p $O; : > $O/counters.txt
one() {   # name, --only pattern, mode
  rm -rf $O/$1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/$1 -- python3 tools/conv_bench.py --only "$2" --mode $3 --img --iters 5 > $O/$1.log 2>&1 || { echo "$1: profiler run failed" | tee -a $O/counters.txt; return; }
  python3 - "$O/$1" "$1" >> $O/counters.txt <<'PY'
import csv, glob, sys, collections, re
d, name = sys.argv[1], sys.argv[2]
f = glob.glob(d + '/*/*_counter_collection.csv'); tr = glob.glob(d + '/*/*_kernel_trace.csv')
def short(n):
    m = re.search(r'(fx16_conv_kernel|fx_conv_kernel|fx_wgrad_kernel)<([^>]*)>', n)
    return (m.group(1) + '<' + m.group(2) + '>') if m else None
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = short(r['Kernel_Name'])
    if k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(tr[0])):
    k = short(r['Kernel_Name'])
    if k: dur[k].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
for k in sorted(agg, key=lambda q: -sum(dur[q])):
    m = {c: sum(v) / len(v) for c, v in agg[k].items()}
    us = sum(dur[k]) / len(dur[k]); gui = m.get('GRBM_GUI_ACTIVE', 0) / 8
    print('%-16s %-44s %7.1f us  clock %.2f GHz  MFMA busy %3.0f%% of active cycles  LDS bank-conflict cycles / LDS active cycles %.3f  (n=%d)'
          % (name, k, us, gui / us / 1e3, 100 * m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / max(gui, 1), m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_LDS_IDX_ACTIVE', 1), 1), len(dur[k])))
PY
}
one fwd_c512_3x3 "c512 h16 k512 3x3 s1 d1" fwd
one fwd_c64_3x3 "c64 h64 k64 3x3" fwd
one fwd_regressor "c2048 h16 k272" fwd
one dgrad_c64_3x3 "c64 h64 k64 3x3" dgrad
one wgrad_c64_3x3 "c64 h64 k64 3x3" wgrad
one wgrad_c512_3x3 "c512 h16 k512 3x3 s1 d1" wgrad
cat $O/counters.txt

#!/bin/bash
# tools/pmc.sh "<conv_bench --only pattern>" <mode> [extra conv_bench flags] : three rocprofv3 --pmc passes over one conv shape; prints per-launch means per kernel instance
sh="$1"; mode="${2:-fwd}"; extra="$3"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES"
P3="TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCC_HIT TCC_MISS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1)); rm -rf gpurun_out/pmcx_$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d gpurun_out/pmcx_$i -- python3 tools/conv_bench.py --only "$sh" --mode $mode --iters 3 $extra > gpurun_out/pmcx_$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, re
def short(n):
    m = re.search(r'(fx_conv_kernel|fx_wgrad_kernel|igemm_kernel)<([^>]*)>', n)
    return (m.group(1) + '<' + m.group(2) + '>') if m else None
for i in (1,2,3):
    f=glob.glob('gpurun_out/pmcx_%d/*/*_counter_collection.csv'%i)
    if not f: print('pass',i,'missing'); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = short(r['Kernel_Name'])
        if k: agg[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k, c), v in sorted(agg.items()): print('%-44s %-28s %.4g'%(k, c, sum(v)/len(v)))
    tr=glob.glob('gpurun_out/pmcx_%d/*/*_kernel_trace.csv'%i)[0]
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(tr)):
        k = short(r['Kernel_Name'])
        if k: d[k].append((float(r['End_Timestamp'])-float(r['Start_Timestamp']))/1e3)
    for k, v in sorted(d.items()): print('  %-44s us/launch %.1f (n=%d)'%(k, sum(v)/len(v),len(v)))
PY

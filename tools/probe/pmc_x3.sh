#!/bin/bash
# Two rocprofv3 --pmc passes over the bf16x3 probe GEMM (one shape, 3 launches): where do the cycles of the six-product inner loop go?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > gpurun_out/_x3_once.py <<'PY'
import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'tools/probe/bf16x3_gemm.so'))
lib.bf16x3_gemm.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
m, n, k = 2048, 16384, 1024
A = torch.randn(m, k, device='cuda'); B = torch.randn(n, k, device='cuda'); C = torch.empty(m, n, device='cuda')
for _ in range(3):
    assert lib.bf16x3_gemm(A.data_ptr(), B.data_ptr(), C.data_ptr(), m, n, k, 6, None) == 0
torch.cuda.synchronize()
PY
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf gpurun_out/pmcx3_$i
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d gpurun_out/pmcx3_$i -- python3 gpurun_out/_x3_once.py > gpurun_out/pmcx3_$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for i in (1, 2):
    f = glob.glob('gpurun_out/pmcx3_%d/*/*_counter_collection.csv' % i)
    if not f: print('pass', i, 'missing'); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if 'bf16x3' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items(): print('%-28s %.4g' % (k, sum(v) / len(v)))
    tr = glob.glob('gpurun_out/pmcx3_%d/*/*_kernel_trace.csv' % i)[0]
    d = [(float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(tr)) if 'bf16x3' in r['Kernel_Name']]
    print('  us/launch %.1f (n=%d)' % (sum(d) / len(d), len(d)))
PY

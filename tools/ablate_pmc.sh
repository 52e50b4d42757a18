#!/bin/bash
# clock and matrix-pipe occupancy of the ablation builds (tools/ablate.sh build) on one shape: GRBM_GUI_ACTIVE / duration = clock, MFMA busy cycles / (SIMDs x active cycles) = occupancy
pat="$1"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in product noload nostage read_mfma mfma_only; do
  rm -rf gpurun_out/abl_$n
  lib=$PWD/3d-pose-estimation-with-previleged-information_amd/csrc/libp3d_abl_$n.so
  [ $n = product ] && lib=$PWD/3d-pose-estimation-with-previleged-information_amd/csrc/libp3d_hip.so
  P3D_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/abl_$n -- python3 tools/conv_bench.py --only "$pat" --mode fwd --img --iters 5 > gpurun_out/abl_$n.log 2>&1
  python3 - "$n" <<'PY'
import csv, glob, sys, collections
n = sys.argv[1]
f = glob.glob('gpurun_out/abl_%s/*/*_counter_collection.csv' % n)
tr = glob.glob('gpurun_out/abl_%s/*/*_kernel_trace.csv' % n)
if not f or not tr: print(n, 'missing'); sys.exit()
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if 'fx_conv_kernel<1' in r['Kernel_Name'] or 'fx_conv_kernel<2' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
d = [(float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(tr[0])) if 'fx_conv_kernel<1' in r['Kernel_Name'] or 'fx_conv_kernel<2' in r['Kernel_Name']]
us = sum(d) / len(d); m = {k: sum(v) / len(v) for k, v in agg.items()}
gui = m.get('GRBM_GUI_ACTIVE', 0) / 8
print('%-10s %7.1f us  clock %.2f GHz  MFMA busy %.0f%% of active cycles' % (n, us, gui / us / 1e3, 100 * m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / max(gui, 1)), {k: '%.3g' % v for k, v in m.items()})
PY
done

#!/bin/bash
# HBM-side traffic of the conv kernel: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE cannot share a pass), 3 steps each.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  P3D_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('gpurun_out/pmc_%s/*/*_counter_collection.csv' % c)[0]
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = 'igemm' if 'igemm_kernel' in r['Kernel_Name'] else 'other'
        tot[k] += float(r['Counter_Value']); n[k] += 1
    out[c] = dict(sum_kb=tot['igemm'], launches=n['igemm'], kb_per_launch=tot['igemm'] / max(n['igemm'], 1))
out['bytes_per_launch_raw'] = (out['FETCH_SIZE']['kb_per_launch'] + out['WRITE_SIZE']['kb_per_launch']) * 1024
out['note'] = 'raw rocprofv3 FETCH_SIZE/WRITE_SIZE (KB) of p3d::igemm_kernel over the steps of that run; dword buffer loads, so the x2 FETCH correction for 16-B/lane streams is not applied'
json.dump(out, open('gpurun_out/traffic.json', 'w'), indent=1)
print(json.dumps(out))
PY

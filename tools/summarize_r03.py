#!/usr/bin/env python3
"""gpurun_out/prof_{serial,overlap}/p_kernel_stats.csv + gpurun_out/pmc_{FETCH,WRITE}_SIZE (tools/profile_r03.sh) -> profiles/r03_*:
raw per-kernel CSVs, grouped per-step tables (r03_tables.md) and the HBM-side traffic of the conv kernels (r03_traffic.json)."""
import collections
import csv
import glob
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go = lambda *p: os.path.join(ROOT, 'gpurun_out', *p)
STEPS = 7            # bench.py --steps 5 --warmup 2 --lean


def group(name):
    name = name.replace('void ', '')
    if 'fx_conv_kernel<1' in name:
        return 'p3d::fx_conv_kernel<AMODE 1> (x3 conv fwd / dgrad, image-fed)'
    if 'fx_conv_kernel' in name:
        return 'p3d::fx_conv_kernel<AMODE 0> (x3 conv fwd / dgrad, fp32 activations split in the kernel)'
    if 'fx_wgrad_kernel' in name:
        return 'p3d::fx_wgrad_kernel (x3 conv wgrad, all instances)'
    if 'fx_act_image_kernel<2>' in name:
        return 'p3d::fx_act_image_kernel<2> (BatchNorm-backward map -> gradient image)'
    if 'fx_act_image_kernel<1>' in name:
        return 'p3d::fx_act_image_kernel<1> (BatchNorm + ReLU -> activation image)'
    if 'fx_act_image_kernel<0>' in name:
        return 'p3d::fx_act_image_kernel<0> (fp32 -> image: regressor operands)'
    if 'igemm_kernel' in name:
        return 'p3d::igemm_kernel (fp32-MFMA conv)'
    m = re.match(r'(p3d::\w+)', name)
    if m:
        return m.group(1)
    m = re.match(r'(at::native::\w+)', name)
    return m.group(1) if m else name[:60]


def table(path):
    rows = {}
    for r in csv.DictReader(open(path)):
        e = rows.setdefault(group(r['Name']), [0, 0.0])
        e[0] += int(r['Calls'])
        e[1] += float(r['TotalDurationNs'])
    tot = sum(v[1] for v in rows.values())
    out = ['| kernel | calls/step | ms/step | avg us/launch | share |', '|---|---|---|---|---|']
    for g, (calls, ns) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:24]:
        out.append('| %s | %.1f | %.3f | %.1f | %.1f%% |' % (g, calls / STEPS, ns / STEPS / 1e6, ns / calls / 1e3, 100 * ns / tot))
    out.append('| **all kernels** |  | **%.2f** |  |  |' % (tot / STEPS / 1e6))
    return '\n'.join(out), rows


def main():
    parts = []
    for name, title in (('serial', 'every kernel on one stream (P3D_WGRAD_STREAM=0 P3D_BLOCK_SIDE=0)'), ('overlap', 'product default: weight gradients on the second stream')):
        src = go('prof_%s' % name, 'p_kernel_stats.csv')
        if not os.path.exists(src):
            continue
        shutil.copy(src, os.path.join(ROOT, 'profiles', 'r03_%s_kernel_stats.csv' % name))
        t, rows = table(src)
        line = [l for l in open(go('prof_%s.log' % name)) if l.startswith('{"metric"')]
        parts.append('### `bench.py --steps 5 --warmup 2 --lean`, %s\n\nrocprofv3 --kernel-trace --stats, 7 steps in the process.\n\n%s\n\nbench line of this run (under the profiler): %s' % (title, t, line[-1].strip() if line else 'n/a'))
    open(os.path.join(ROOT, 'profiles', 'r03_tables.md'), 'w').write('\n\n'.join(parts) + '\n')
    out = {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        f = glob.glob(go('pmc_%s' % c, '*', '*_counter_collection.csv')) + glob.glob(go('pmc_%s' % c, '*_counter_collection.csv'))
        # gpurun_out/ keeps files of earlier calls and of helper processes: the run of interest is the newest trace that holds the x3 kernels
        f = sorted((x for x in f if 'fx_conv_kernel' in open(x).read()), key=os.path.getmtime, reverse=True)
        if not f:
            continue
        tot, n = collections.defaultdict(float), collections.defaultdict(int)
        for r in csv.DictReader(open(f[0])):
            k = 'x3' if ('fx_conv_kernel' in r['Kernel_Name'] or 'fx_wgrad_kernel' in r['Kernel_Name']) else ('igemm' if 'igemm_kernel' in r['Kernel_Name'] else 'other')
            tot[k] += float(r['Counter_Value'])
            n[k] += 1
        out[c] = {k: dict(sum_kb=tot[k], launches=n[k], kb_per_launch=tot[k] / max(n[k], 1)) for k in tot}
    if 'FETCH_SIZE' in out and 'WRITE_SIZE' in out:
        f, w = out['FETCH_SIZE']['x3'], out['WRITE_SIZE']['x3']
        out['raw_bytes_per_launch'] = (f['kb_per_launch'] + w['kb_per_launch']) * 1024
        # MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE counts a wide (16 B / lane) coalesced read at half its bytes; the x3 kernels fetch their
        # operands with 16-B buffer loads (the shifted taps of the 3x3 layers with dword loads: uncalibrated, left at the same factor); WRITE_SIZE is exact.
        out['bytes_per_launch'] = (2 * f['kb_per_launch'] + w['kb_per_launch']) * 1024
        out['note'] = ('p3d::fx_conv_kernel + fx_wgrad_kernel launches of `bench.py --steps 2 --warmup 1 --lean`, every kernel on one stream; '
                       'bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction for 16-B/lane reads, MI355X_MICROARCH.md), raw_bytes_per_launch = uncorrected sum')
        json.dump(out, open(os.path.join(ROOT, 'profiles', 'r03_traffic.json'), 'w'), indent=1)
    print(json.dumps({k: v for k, v in out.items() if not isinstance(v, dict)}))


if __name__ == '__main__':
    main()

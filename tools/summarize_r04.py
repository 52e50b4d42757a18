#!/usr/bin/env python3
"""gpurun_out/prof_{serial,overlap}/p_kernel_stats.csv + gpurun_out/pmc_{FETCH,WRITE}_SIZE (tools/profile_r04.sh) -> profiles/r04_*:
raw per-kernel CSVs, grouped per-step tables (r04_tables.md) and the HBM-side traffic of the conv kernels SPLIT by pass (forward / data gradient / weight
gradient) and layer group (stem, layer1 .. layer4, regressor), each against its algorithmic bytes (r04_traffic.json)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go = lambda *p: os.path.join(ROOT, 'gpurun_out', *p)
STEPS = 7            # bench.py --steps 5 --warmup 2 --lean
BATCH = 64


def group(name):
    name = name.replace('void ', '')
    if 'fx_conv_kernel<1' in name or 'fx16_conv_kernel' in name:
        return 'p3d::fx_conv_kernel<AMODE 1> + fx16_conv_kernel (x3 conv fwd / dgrad, image-fed)'
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
    for g, (calls, ns) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:26]:
        out.append('| %s | %.1f | %.3f | %.1f | %.1f%% |' % (g, calls / STEPS, ns / STEPS / 1e6, ns / calls / 1e3, 100 * ns / tot))
    out.append('| **all kernels** |  | **%.2f** |  |  |' % (tot / STEPS / 1e6))
    return '\n'.join(out), rows


# ResNet-50 at 256 x 256, stride 16, in FORWARD launch order: (group, Cin, Hin, Cout, k, stride)
def r50_forward_order():
    seq = [('stem', 3, 256, 64, 7, 2)]
    inpl, h = 64, 64
    for li, (planes, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 1))):
        g = 'layer%d' % (li + 1)
        for b in range(blocks):
            s = stride if b == 0 else 1
            if b == 0:
                seq.append((g, inpl, h, planes * 4, 1, s))           # downsample first? no: the executor launches conv1, conv2, conv3, then the downsample -- order fixed below
            seq.append((g, inpl, h, planes, 1, 1))
            seq.append((g, planes, h, planes, 3, s))
            h2 = h // s
            seq.append((g, planes, h2, planes * 4, 1, 1))
            inpl, h = planes * 4, h2
    seq.append(('regressor', 2048, 16, 272, 3, 1))
    return seq


def alg_bytes(cin, hin, cout, k, stride):
    """fp32 input + output + weights of one pass (the figure SURVEY.md 8(d) prices a launch at; the same for forward, data gradient and weight gradient)"""
    ho = hin // stride
    return 4.0 * (BATCH * cin * hin * hin + BATCH * cout * ho * ho + cout * cin * k * k)


def traffic():
    per = {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        f = glob.glob(go('pmc_%s' % c, '*', '*_counter_collection.csv')) + glob.glob(go('pmc_%s' % c, '*_counter_collection.csv'))
        f = sorted((x for x in f if 'fx_conv_kernel' in open(x).read()), key=os.path.getmtime, reverse=True)
        if not f:
            return None
        rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Dispatch_Id']))
        # cut into steps at adam_kernel; inside a step everything before the loss kernel is the forward pass
        steps, cur = [], []
        for r in rows:
            cur.append(r)
            if 'adam_kernel' in r['Kernel_Name']:
                steps.append(cur)
                cur = []
        per[c] = steps[-1]           # the last complete step
    order = r50_forward_order()
    result = collections.defaultdict(lambda: dict(FETCH_SIZE=0.0, WRITE_SIZE=0.0, launches=0, alg=0.0))
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        step = per[c]
        loss_at = next(i for i, r in enumerate(step) if 'pose_loss' in r['Kernel_Name'] or 'softargmax' in r['Kernel_Name'])
        is_conv = lambda n: 'fx_conv_kernel' in n or 'fx16_conv_kernel' in n
        fwd = [r for r in step[:loss_at] if is_conv(r['Kernel_Name'])]
        dgr = [r for r in step[loss_at:] if is_conv(r['Kernel_Name'])]
        wgr = [r for r in step[loss_at:] if 'fx_wgrad_kernel' in r['Kernel_Name']]
        assert len(fwd) == 54 and len(dgr) == 53 and len(wgr) == 54, (len(fwd), len(dgr), len(wgr))
        # forward launches follow module order (within a block: conv1, conv2, conv3, downsample -- the byte totals per layer GROUP do not depend on that order);
        # backward launches run the groups in reverse
        groups_f = [o[0] for o in order]
        groups_b = list(reversed(groups_f))
        alg_group = collections.defaultdict(float)
        for o in order:
            alg_group[o[0]] += alg_bytes(*o[1:])
        for cls, lst, groups in (('fwd', fwd, groups_f), ('dgrad', dgr, [g for g in groups_b if g != 'stem'] if len(dgr) == 53 else groups_b), ('wgrad', wgr, groups_b)):
            for r, g in zip(lst, groups):
                e = result[(cls, g)]
                e[c] += float(r['Counter_Value']) * 1024.0
                if c == 'FETCH_SIZE':
                    e['launches'] += 1
            for g in set(groups):
                result[(cls, g)]['alg'] = alg_group[g]
    out = {'by_pass_and_group': {}, 'by_pass': {}}
    tot = dict(f=0.0, w=0.0, alg=0.0, n=0)
    bypass = collections.defaultdict(lambda: dict(f=0.0, w=0.0, alg=0.0, n=0))
    for (cls, g), e in sorted(result.items()):
        meas = 2 * e['FETCH_SIZE'] + e['WRITE_SIZE']
        out['by_pass_and_group']['%s %s' % (cls, g)] = dict(launches=e['launches'], fetch_raw_mb=round(e['FETCH_SIZE'] / 1e6, 1), write_mb=round(e['WRITE_SIZE'] / 1e6, 1),
                                                           measured_mb=round(meas / 1e6, 1), algorithmic_mb=round(e['alg'] / 1e6, 1), ratio=round(meas / e['alg'], 2))
        for d in (tot, bypass[cls]):
            d['f'] += e['FETCH_SIZE']; d['w'] += e['WRITE_SIZE']; d['alg'] += e['alg']; d['n'] += e['launches']
    for cls, d in bypass.items():
        out['by_pass'][cls] = dict(launches=d['n'], measured_mb=round((2 * d['f'] + d['w']) / 1e6, 1), algorithmic_mb=round(d['alg'] / 1e6, 1), ratio=round((2 * d['f'] + d['w']) / d['alg'], 2))
    out['raw_bytes_per_launch'] = (tot['f'] + tot['w']) / tot['n']
    out['bytes_per_launch'] = (2 * tot['f'] + tot['w']) / tot['n']
    out['algorithmic_bytes_per_launch'] = tot['alg'] / tot['n']
    out['launches_per_step'] = tot['n']
    try:
        out['commit'] = subprocess.run(['git', '-C', ROOT, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        out['commit'] = None
    out['note'] = ('conv launches (fx_conv_kernel / fx16_conv_kernel / fx_wgrad_kernel) of ONE step of `bench.py --steps 2 --warmup 1 --lean`, every kernel on one stream, separate '
                   '--pmc FETCH_SIZE and --pmc WRITE_SIZE passes; measured = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts 16-B/lane reads at half, MI355X_MICROARCH.md); '
                   'algorithmic = fp32 input + output + weights of the pass; split-K / slab reduce launches are not in these rows')
    return out


def main():
    parts = []
    for name, title in (('serial', 'every kernel on one stream (P3D_WGRAD_STREAM=0 P3D_BLOCK_SIDE=0)'), ('overlap', 'product default: weight gradients on the second stream')):
        src = go('prof_%s' % name, 'p_kernel_stats.csv')
        if not os.path.exists(src):
            continue
        shutil.copy(src, os.path.join(ROOT, 'profiles', 'r04_%s_kernel_stats.csv' % name))
        t, rows = table(src)
        line = [l for l in open(go('prof_%s.log' % name)) if l.startswith('{"metric"')]
        parts.append('### `bench.py --steps 5 --warmup 2 --lean`, %s\n\nrocprofv3 --kernel-trace --stats, 7 steps in the process.\n\n%s\n\nbench line of this run (under the profiler): %s' % (title, t, line[-1].strip() if line else 'n/a'))
    open(os.path.join(ROOT, 'profiles', 'r04_tables.md'), 'w').write('\n\n'.join(parts) + '\n')
    out = traffic()
    if out:
        json.dump(out, open(os.path.join(ROOT, 'profiles', 'r04_traffic.json'), 'w'), indent=1)
        print(json.dumps(out['by_pass'], indent=1))
        for k, v in out['by_pass_and_group'].items():
            print('%-18s %s' % (k, v))
        print(json.dumps({k: v for k, v in out.items() if not isinstance(v, dict)}))


if __name__ == '__main__':
    main()

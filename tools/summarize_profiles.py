#!/usr/bin/env python3
"""Turn gpurun_out/prof_{serial,overlap,half}/p_kernel_stats.csv (tools/profile_round.sh) into profiles/rNN_* files:
the raw per-kernel CSVs, a grouped markdown table per configuration and the bench line printed by the same command."""
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else 'r01'
STEPS = 13            # bench.py --steps 5 --warmup 2: 2 + 5 timed + 1 + 5 kernel-pass steps


def group(name):
    name = name.replace('void ', '')
    if 'igemm_kernel' in name:
        return 'p3d::igemm_kernel (fp32 conv fwd/dgrad/wgrad, all template instances)'
    m = re.match(r'(p3d::\w+)', name)
    if m:
        return m.group(1)
    m = re.match(r'(at::native::\w+).*?(CUDAFunctor\w*|FillFunctor|MulFunctor)?', name)
    return (m.group(1) + (' ' + m.group(2) if m and m.group(2) else '')) if m else name[:60]


def table(path):
    rows = {}
    for r in csv.DictReader(open(path)):
        g = group(r['Name'])
        e = rows.setdefault(g, [0, 0.0])
        e[0] += int(r['Calls'])
        e[1] += float(r['TotalDurationNs'])
    tot = sum(v[1] for v in rows.values())
    out = ['| kernel | calls/step | ms/step | avg us/launch | share |', '|---|---|---|---|---|']
    for g, (calls, ns) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:18]:
        out.append('| %s | %.1f | %.3f | %.1f | %.1f%% |' % (g, calls / STEPS, ns / STEPS / 1e6, ns / calls / 1e3, 100 * ns / tot))
    out.append('| **all kernels** |  | **%.2f** |  |  |' % (tot / STEPS / 1e6))
    return '\n'.join(out)


def bench_line(log):
    for line in open(log):
        if line.startswith('{"metric"'):
            return json.loads(line)
    return None


def main():
    parts = []
    for cfg, title in (('serial', 'P3D_WGRAD_STREAM=0 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline  (every kernel alone on the GPU)'),
                       ('overlap', 'python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline  (product default: wgrad kernels on a second stream)'),
                       ('half', 'python3 bench.py --half --steps 5 --warmup 2 --no-cpu-baseline  (-half_acc: fp16 NHWC path, informational)')):
        src = os.path.join(ROOT, 'gpurun_out', 'prof_%s' % cfg, 'p_kernel_stats.csv')
        if not os.path.exists(src):
            continue
        shutil.copy(src, os.path.join(ROOT, 'profiles', '%s_%s_kernel_stats.csv' % (RND, cfg)))
        b = bench_line(os.path.join(ROOT, 'gpurun_out', 'prof_%s.log' % cfg))
        parts.append('### `%s`\n\nrocprofv3 --kernel-trace --stats, %d steps in the process (2 warm-up + 5 timed + 1 + 5 kernel-pass).\n\n%s\n' % (title, STEPS, table(src)))
        if b:
            r = b['roofline']
            parts.append('bench line of this run (under the profiler): %.1f crops/s, %.2f ms/step; roofline.achieved %.1f TFLOP/s (conv ms/step %s), '
                         'achieved_in_timed_region %s\n' % (b['value'], b['ms_per_step'], r['achieved'], json.dumps(r['conv_ms_per_step']), r['achieved_in_timed_region']))
    tj = os.path.join(ROOT, 'gpurun_out', 'traffic.json')
    if os.path.exists(tj):
        shutil.copy(tj, os.path.join(ROOT, 'profiles', '%s_traffic.json' % RND))
    open(os.path.join(ROOT, 'gpurun_out', '%s_tables.md' % RND), 'w').write('\n'.join(parts))
    print('\n'.join(parts))


if __name__ == '__main__':
    main()

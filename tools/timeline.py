#!/usr/bin/env python3
"""Timeline of one steady-state training step from a rocprofv3 --kernel-trace CSV: per hardware queue the busy time and the gaps, the largest gaps of the
launch queue with the kernels either side, and the launch queue's time by kernel class.  A step ends at p3d::adam_kernel.
usage: python tools/timeline.py <kernel_trace.csv> [step index from the end, default 2]"""
import collections, csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')) for r in rows), key=lambda k: k[0])
adams = [i for i, k in enumerate(ks) if 'adam_kernel' in k[2] or 'adam_dev_kernel' in k[2]]
if len(adams) < back + 1:
    sys.exit('not enough steps in the trace')
lo, hi = adams[-back - 1] + 1, adams[-back] + 1
step = ks[lo:hi]
t0, t1 = step[0][0], max(k[1] for k in step)
print('step: %d kernels, %.3f ms from first start to last end' % (len(step), (t1 - t0) / 1e6))


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    m = re.match(r'(?:p3d::)?(\w+)(<[^>]*>)?', name)
    return (m.group(1) + (m.group(2) or '')) if m else name[:40]


byq = collections.defaultdict(list)
for k in step:
    byq[k[3]].append(k)
main = max(byq, key=lambda q: len(byq[q]))
for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(b - a for a, b, _, _ in lst)
    print('queue %s%s: %d kernels, busy %.3f ms, span %.3f ms' % (q, ' (launch stream)' if q == main else '', len(lst), busy / 1e6, (max(b for _, b, _, _ in lst) - lst[0][0]) / 1e6))
# union busy over all queues
ev = sorted([(a, 1) for a, b, _, _ in step] + [(b, -1) for a, b, _, _ in step])
depth, last, idle = 0, t0, 0
for t, d in ev:
    if depth == 0:
        idle += t - last
    depth += d
    last = t
print('no kernel running on any queue: %.3f ms' % (idle / 1e6))
lst = byq[main]
gaps = sorted(((b[0] - a[1], short(a[2]), short(b[2])) for a, b in zip(lst, lst[1:]) if b[0] > a[1]), reverse=True)
print('launch-stream gaps: total %.3f ms in %d gaps; > 10 us: %.3f ms' % (sum(g[0] for g in gaps) / 1e6, len(gaps), sum(g[0] for g in gaps if g[0] > 10000) / 1e6))
for g in gaps[:12]:
    print('   %7.1f us between %s and %s' % (g[0] / 1e3, g[1], g[2]))
for q in byq:
    cls = collections.Counter()
    cnt = collections.Counter()
    for a, b, n, _ in byq[q]:
        cls[short(n)] += b - a
        cnt[short(n)] += 1
    print('queue %s by kernel:' % q)
    for n, t in cls.most_common(14):
        print('   %-44s %4d  %8.3f ms' % (n[:44], cnt[n], t / 1e6))

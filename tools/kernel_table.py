"""Per-kernel table (calls / step, ms / step, average us) from a rocprofv3 --kernel-trace --stats csv directory.  usage: kernel_table.py DIR STEPS"""
import csv, glob, os, re, sys

d, steps = sys.argv[1], float(sys.argv[2])
path = [p for p in glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True)][0]
rows = list(csv.DictReader(open(path)))
tot = 0.0
out = []
for r in rows:
    name = r['Name']
    name = re.sub(r'\(.*', '', name)
    name = name.replace('void ', '').replace('p3d::', '')
    ns = float(r['TotalDurationNs'])
    tot += ns
    out.append((ns, int(r['Calls']), name))
out.sort(reverse=True)
print('%-64s %10s %10s %10s %7s' % ('kernel', 'calls/step', 'ms/step', 'avg us', 'share'))
for ns, calls, name in out[:40]:
    print('%-64s %10.1f %10.3f %10.1f %6.1f%%' % (name[:64], calls / steps, ns / 1e6 / steps, ns / 1e3 / calls, 100 * ns / tot))
print('%-64s %10s %10.3f' % ('all kernels', '', tot / 1e6 / steps))

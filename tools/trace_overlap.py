"""Read a rocprofv3 kernel-trace CSV and report kernels of ONE queue that overlap in time (in-order streams must not)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
byq = collections.defaultdict(list)
for r in rows:
    byq[(r.get('Queue_Id'), r.get('Stream_Id', ''))].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]))
for q, ks in byq.items():
    ks.sort()
    over = 0
    ex = []
    for a, b in zip(ks, ks[1:]):
        if b[0] < a[1]:
            over += 1
            if len(ex) < 6:
                ex.append((a[2], b[2], a[1] - b[0]))
    print('queue', q, 'kernels', len(ks), 'overlapping consecutive pairs', over)
    for e in ex:
        print('    ', e)

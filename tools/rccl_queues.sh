# which HSA queue every kernel of the step was dispatched to, with and without the RCCL group (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for cfg in none late first; do
  case $cfg in none) unset P3D_FORCE_DIST P3D_BENCH_JOIN_FIRST;; late) export P3D_FORCE_DIST=1; unset P3D_BENCH_JOIN_FIRST;; first) export P3D_FORCE_DIST=1 P3D_BENCH_JOIN_FIRST=1;; esac
  rm -rf gpurun_out/q_$cfg
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/q_$cfg -o q -- python3 bench.py --steps 5 --warmup 3 --lean > gpurun_out/q_$cfg.log 2>&1
  tail -1 gpurun_out/q_$cfg.log | cut -c1-200
  python3 - $cfg <<'PY'
import csv, glob, sys, collections
cfg = sys.argv[1]
f = glob.glob('gpurun_out/q_%s/**/*kernel_trace.csv' % cfg, recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) // 2:]                      # the later (timed) steps
q = collections.defaultdict(lambda: [0, 0, collections.Counter()])
for r in rows:
    e = q[r['Queue_Id']]
    e[0] += 1; e[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); e[2][r['Kernel_Name'][:40]] += 1
span = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])
for k, (n, ns, names) in sorted(q.items()):
    print('  %s queue %s: %5d kernels, busy %.1f%% of span | %s' % (cfg, k, n, 100.0 * ns / span, ', '.join('%s x%d' % kv for kv in names.most_common(4))))
PY
done

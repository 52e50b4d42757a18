# VERDICT r01 item 7, after the fix: the weight-gradient stream is chosen by a concurrency probe; P3D_SIDE_STREAM=torch is the old behaviour (PyTorch pool stream)
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
b() { python bench.py --lean --steps 30 --warmup 10 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
for prio in probe torch; do
  export P3D_SIDE_STREAM=$prio
  echo "side=$prio  no group     $(b)"
  echo "side=$prio  group late   $(P3D_FORCE_DIST=1 b)"
  echo "side=$prio  group first  $(P3D_FORCE_DIST=1 P3D_BENCH_JOIN_FIRST=1 b)"
done

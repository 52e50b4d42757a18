# VERDICT r01 item 7 at step level: the contract step (lean bench, 30 timed steps) with no group, with a single-rank RCCL group joined after the
# allocations (bench.py's order) and with the group joined first; twice each, interleaved, in one box
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for rep in 1 2; do
  echo "none   $(python bench.py --lean --steps 30 --warmup 10 2>/dev/null | tail -1)"
  echo "late   $(P3D_FORCE_DIST=1 python bench.py --lean --steps 30 --warmup 10 2>/dev/null | tail -1)"
  echo "first  $(P3D_FORCE_DIST=1 P3D_BENCH_JOIN_FIRST=1 python bench.py --lean --steps 30 --warmup 10 2>/dev/null | tail -1)"
done

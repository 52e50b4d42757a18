b() { python bench.py --lean --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
for i in 1 2; do
echo "partial_depthnet r50 bs64  x3-masked   $(b --family partial_depthnet)"
echo "partial_depthnet r50 bs64  fp32-masked $(P3D_FX_MASKED=0 b --family partial_depthnet)"
done
echo "partial_fusionnet bs32     x3-masked   $(b --family partial_fusionnet --batch 32)"
echo "partial_fusionnet bs32     fp32-masked $(P3D_FX_MASKED=0 b --family partial_fusionnet --batch 32)"
python bench.py --family partial_depthnet --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['conv_paths'], d['roofline']['conv_ms_per_step'])"

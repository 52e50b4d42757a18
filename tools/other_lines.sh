# informational lines of the other BASELINE configurations (lean bench: 20 timed steps after 5)
b() { python bench.py --lean --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 crops\/s  \2 ms/'; }
echo "depthnet r50 bs64            $(b)"
echo "partial_depthnet r50 bs64    $(b --family partial_depthnet)"
echo "fusionnet r50 bs32           $(b --family fusionnet --batch 32)"
echo "fusionnet r50 bs32 augment   $(b --family fusionnet --batch 32 --augment)"
echo "partial_fusionnet r50 bs32   $(b --family partial_fusionnet --batch 32)"
echo "depthnet r18 bs64            $(b --model resnet18)"
echo "depthnet r18 bs8             $(b --model resnet18 --batch 8)"
echo "depthnet r50 bs64 half       $(b --half)"
echo "depthnet r50 bs64 P3D_X3=0   $(P3D_X3=0 b)"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1

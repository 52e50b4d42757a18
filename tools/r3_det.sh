cfg="resnet18 4 256"
run() { echo "== $1"; for i in 1 2 3 4 5 6 7 8; do env $2 DET_REPS=8 python tools/debug_det.py $cfg 2>&1 | grep -v "amdgpu.ids\|^done" | sed "s/^/  try $i: /"; done; }
run "default (two streams)" "X=1"
run "P3D_BLOCK_SIDE=0 (only the per-layer wgrads on the side stream)" "P3D_BLOCK_SIDE=0"
run "P3D_SIDE_STREAM=torch" "P3D_SIDE_STREAM=torch"
run "no hooks on the stem" "DET_HOOKS=0"
run "P3D_WGRAD_STREAM=0" "P3D_WGRAD_STREAM=0"

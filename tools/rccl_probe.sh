# the probe under the allocator / registration knobs VERDICT r01 item 7 names
run() { echo "== $*"; env "$@" python tools/rccl_alloc_probe.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn"; }
run P3D_NOOP=1
run PYTORCH_HIP_ALLOC_CONF=expandable_segments:True
run TORCH_NCCL_USE_TENSOR_REGISTER_ALLOCATOR_HOOK=0 NCCL_DMABUF_ENABLE=0
run RCCL_MSCCL_ENABLE=0 RCCL_MSCCLPP_ENABLE=0
run HSA_ENABLE_SDMA=0

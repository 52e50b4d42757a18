#!/usr/bin/env python3
"""is device memory that is first allocated after an RCCL communicator exists slower for the kernels, and why?
One process, one rank (the nccl backend of torch.distributed IS RCCL here).  Tensors of the same size are allocated BEFORE and AFTER
init_process_group; for each we print what the HIP runtime says about the allocation (hipPointerGetAttributes, hipMemGetAddressRange) and time
an HBM-bound pass (copy) and the MFMA-bound x3 conv forward of a layer3 1x1 on it.  `--late-join` joins the group only after everything is
allocated (the order bench.py uses); the env knobs under test are taken from the environment as they are.  GPU box only."""
import argparse
import ctypes
import importlib
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('3d-pose-estimation-with-previleged-information_amd')
ops, L = pkg.ops, pkg._lib.lib()
hip = ctypes.CDLL('libamdhip64.so')


class Attr(ctypes.Structure):
    _fields_ = [('type', ctypes.c_int), ('device', ctypes.c_int), ('devicePointer', ctypes.c_void_p), ('hostPointer', ctypes.c_void_p),
                ('isManaged', ctypes.c_int), ('allocationFlags', ctypes.c_uint)]


def describe(t):
    a = Attr()
    rc = hip.hipPointerGetAttributes(ctypes.byref(a), ctypes.c_void_p(t.data_ptr()))
    base, size = ctypes.c_void_p(), ctypes.c_size_t()
    rc2 = hip.hipMemGetAddressRange(ctypes.byref(base), ctypes.byref(size), ctypes.c_void_p(t.data_ptr()))
    return dict(rc=rc, type=a.type, managed=a.isManaged, flags=a.allocationFlags, rc_range=rc2, base=hex(base.value or 0), range_mb=round(size.value / 2 ** 20, 1),
                ptr=hex(t.data_ptr()))


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def make():
    x = torch.randn(64, 1024, 16, 16, device='cuda')
    w = torch.randn(2048, 1024, 1, 1, device='cuda') * 0.03
    y = torch.empty(64, 2048, 16, 16, device='cuda')
    big = torch.randn(64 * 2 ** 20, device='cuda')        # 256 MB
    big2 = torch.empty_like(big)
    return x, w, y, big, big2


def measure(tag, ts):
    x, w, y, big, big2 = ts
    d = ops._desc(x.shape, w.shape, 1, 0, 1)
    ws = torch.empty(max(L.p3d_conv2d_fwd_workspace_bytes(ctypes.byref(d)), 16), dtype=torch.uint8, device='cuda')
    p, st = ops._p, ops._stream()
    conv = timeit(lambda: L.p3d_conv2d_fwd(ctypes.byref(d), p(x), p(w), None, None, None, p(y), p(ws), ws.numel(), st))
    copy = timeit(lambda: big2.copy_(big))
    print('%-28s conv %.1f us  copy256MB %.1f us (%.2f TB/s)  x:%s' % (tag, conv, copy, 2 * big.numel() * 4 / copy / 1e6, describe(x)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--late-join', action='store_true')
    a = ap.parse_args()
    torch.cuda.set_device(0)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    before = make()
    measure('allocated, no group yet', before)
    t0 = time.time()
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    g = torch.ones(1, device='cuda')
    dist.all_reduce(g)                       # the communicator really exists now
    torch.cuda.synchronize()
    print('group joined in %.1f s' % (time.time() - t0), flush=True)
    measure('old tensors, group exists', before)
    if not a.late_join:
        after = make()
        measure('NEW tensors, group exists', after)
        measure('old tensors again', before)
    dist.destroy_process_group()
    measure('old tensors, group destroyed', before)


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Per-shape sweep of the split count of the x3 conv launches (p3d_fx_tune): which split-K / slab count is fastest for each ResNet-50 layer class.
GPU box only; prints one line per (shape, mode) with the time at every split count tried and the built-in plan's time."""
import ctypes
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = sys.argv[:1]
import tools.conv_bench as cb       # noqa: E402  (shape list, timeit)

pkg = importlib.import_module('3d-pose-estimation-with-previleged-information_amd')
ops, L = pkg.ops, pkg._lib.lib()


def main():
    batch = 64
    for (c, h, k, ks, st, dil, cnt) in cb.R50:
        if c < 128 or k < 128:
            continue
        pad = dil * (ks - 1) // 2
        x = torch.randn(batch, c, h, h, device='cuda')
        w = torch.randn(k, c, ks, ks, device='cuda') * 0.05
        d = ops._desc(x.shape, w.shape, st, pad, dil)
        y = torch.empty(batch, k, d.Ho, d.Wo, device='cuda')
        dy, dx, dw = torch.randn_like(y), torch.empty_like(x), torch.empty_like(w)
        p, s_ = ops._p, ops._stream()
        tag = 'c%d h%d k%d %dx%d s%d d%d x%d' % (c, h, k, ks, ks, st, dil, cnt)
        for mode, what, cands in (('wgrad', 2, (0, 256, 384, 512, 576, 640, 768, 1024, 1536, 2048, 3072, 4096)),):
            res = []
            for n in cands:
                L.p3d_fx_tune(what, n)
                if mode == 'fwd':
                    ws = torch.empty(max(L.p3d_conv2d_fwd_workspace_bytes(ctypes.byref(d)), 16), dtype=torch.uint8, device='cuda')
                    fn = lambda: L.p3d_conv2d_fwd(ctypes.byref(d), p(x), p(w), None, None, None, p(y), p(ws), ws.numel(), s_)
                elif mode == 'dgrad':
                    ws = torch.empty(max(L.p3d_conv2d_dgrad_workspace_bytes(ctypes.byref(d)), 16), dtype=torch.uint8, device='cuda')
                    fn = lambda: L.p3d_conv2d_dgrad(ctypes.byref(d), p(dy), p(w), None, None, p(dx), p(ws), ws.numel(), s_)
                else:
                    ws = torch.empty(max(L.p3d_conv2d_wgrad_workspace_bytes(ctypes.byref(d)), 16), dtype=torch.uint8, device='cuda')
                    fn = lambda: L.p3d_conv2d_wgrad(ctypes.byref(d), p(dy), p(x), None, None, p(dw), p(ws), ws.numel(), s_)
                rc = fn()
                if rc != 0:
                    res.append((n, float('nan')))
                    continue
                res.append((n, cb.timeit(fn, 10) * 1e3))
                del ws
            L.p3d_fx_tune(what, 0)
            best = min(res[1:], key=lambda r: r[1])
            print('%-30s %-5s plan %7.1f us | best n=%-2d %7.1f us | %s' % (tag, mode, res[0][1], best[0], best[1], ' '.join('%d:%.0f' % r for r in res[1:])), flush=True)


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Per-shape sweep of the split plans of the image-fed x3 launches (p3d_fx_tune): the slab count of the weight gradient (target block counts) and the
split-K count of forward / data gradient, for each ResNet-50 layer class at batch 64.  GPU box only; one line per (shape, pass)."""
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
WGRAD_TARGETS = (0, 256, 384, 512, 576, 640, 720, 768, 896, 1024, 1280, 1536, 2048, 3072)
CONV_SPLITS = (0, 1, 2, 3, 4, 6, 8)


def main():
    batch = 64
    for (c, h, k, ks, st, dil, cnt) in cb.R50:
        if c % 16 or k % 16:
            continue
        pad = dil * (ks - 1) // 2
        x = torch.randn(batch, c, h, h, device='cuda')
        w = torch.randn(k, c, ks, ks, device='cuda') * 0.05
        d = ops._desc(x.shape, w.shape, st, pad, dil)
        y = torch.empty(batch, k, d.Ho, d.Wo, device='cuda')
        dy, dx, dw = torch.randn_like(y), torch.empty_like(x), torch.empty_like(w)
        p, s_ = ops._p, ops._stream()
        fb, bb = ctypes.c_size_t(), ctypes.c_size_t()
        L.p3d_fx_weight_image_bytes(k, c, ks * ks, ctypes.byref(fb), ctypes.byref(bb))
        wf, wb = torch.empty(fb.value, dtype=torch.uint8, device='cuda'), torch.empty(bb.value, dtype=torch.uint8, device='cuda')
        L.p3d_fx_weight_images(p(w), k, c, ks * ks, p(wf), p(wb), s_)
        xi, dyi = ops.act_image(x), ops.act_image(dy)
        tiles_w = ((k + 127) // 128) * ((c + 127) // 128) * ks * ks
        tag = 'c%d h%d k%d %dx%d s%d d%d x%d' % (c, h, k, ks, ks, st, dil, cnt)
        for mode, what, cands in (('wgrad', 2, WGRAD_TARGETS), ('fwd', 1, CONV_SPLITS), ('dgrad', 1, CONV_SPLITS)):
            res = []
            for n in cands:
                L.p3d_fx_tune(what, n)
                which = {'fwd': 0, 'dgrad': 1, 'wgrad': 2}[mode]
                ws = torch.empty(max(L.p3d_fx_conv_img_workspace_bytes(ctypes.byref(d), which), 16), dtype=torch.uint8, device='cuda')
                if mode == 'fwd':
                    fn = lambda: L.p3d_fx_conv_fwd_img(ctypes.byref(d), p(xi), p(w), p(wf), None, p(y), p(ws), ws.numel(), s_)
                elif mode == 'dgrad':
                    fn = lambda: L.p3d_fx_conv_dgrad_img(ctypes.byref(d), p(dyi), p(w), p(wb), p(dx), p(ws), ws.numel(), s_)
                else:
                    fn = lambda: L.p3d_fx_conv_wgrad_img(ctypes.byref(d), p(dyi), None, p(xi), p(dw), p(ws), ws.numel(), s_)
                rc = fn()
                res.append((n, cb.timeit(fn, 12) * 1e3 if rc == 0 else float('nan')))
                del ws
            L.p3d_fx_tune(what, 0)
            best = min(res[1:], key=lambda r: r[1])
            extra = 'tiles %d' % tiles_w if mode == 'wgrad' else ''
            print('%-30s %-5s plan %7.1f us | best n=%-4d %7.1f us (%+.0f%%) | %s  %s' % (tag, mode, res[0][1], best[0], best[1], 100 * (res[0][1] / best[1] - 1),
                                                                                    ' '.join('%d:%.0f' % r for r in res[1:]), extra), flush=True)


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Per-shape timing of the fp16 NHWC conv kernels (fwd / dgrad / wgrad) over the ResNet-50 layer classes at batch 64. GPU box only."""
import argparse
import ctypes
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('3d-pose-estimation-with-previleged-information_amd')
ops = pkg.ops
L = pkg._lib.lib()
from tools.conv_bench import R50, timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    tot = dict(fwd=0.0, dgrad=0.0, wgrad=0.0)
    totf = 0.0
    print('%-34s %8s | %8s %6s | %8s %6s | %8s %6s' % ('shape', 'GFLOP', 'fwd ms', 'TF', 'dgrad ms', 'TF', 'wgrad ms', 'TF'))
    for (c, h, k, ks, st, dil, cnt) in R50:
        tag = 'c%d h%d k%d %dx%d s%d d%d x%d' % (c, h, k, ks, ks, st, dil, cnt)
        if a.only and a.only not in tag:
            continue
        cp = (c + 7) // 8 * 8
        pad = dil * (ks - 1) // 2
        d = ops._desc((a.batch, cp, h, h), (k, cp, ks, ks), st, pad, dil)
        x = torch.randn(a.batch, h, h, cp, device='cuda').half()
        w = (torch.randn(k, ks, ks, cp, device='cuda') * 0.05).half()
        wt = (torch.randn(cp, ks, ks, k, device='cuda') * 0.05).half()
        y = torch.empty(a.batch, d.Ho, d.Wo, k, device='cuda', dtype=torch.float16)
        dy = torch.randn(a.batch, d.Ho, d.Wo, k, device='cuda').half()
        dx = torch.empty_like(x)
        dw = torch.zeros(k, c, ks, ks, device='cuda')
        ws = torch.empty(max(L.p3d_hconv2d_wgrad_workspace_bytes(ctypes.byref(d)), 16), dtype=torch.uint8, device='cuda')
        st_ = ops._stream()
        p = ops._p
        gflop = 2.0 * a.batch * k * d.Ho * d.Wo * c * ks * ks / 1e9
        t_f = timeit(lambda: L.p3d_hconv2d_fwd(ctypes.byref(d), p(x), p(w), None, None, None, p(y), st_), a.iters)
        t_d = timeit(lambda: L.p3d_hconv2d_dgrad(ctypes.byref(d), p(dy), p(wt), None, p(dx), st_), a.iters) if c > 4 else 0.0
        t_w = timeit(lambda: L.p3d_hconv2d_wgrad(ctypes.byref(d), p(dy), p(x), None, p(dw), c, 1.0, p(ws), ws.numel(), st_), a.iters)
        print('%-34s %8.1f | %8.3f %6.0f | %8.3f %6.0f | %8.3f %6.0f' % (tag, gflop, t_f, gflop / t_f, t_d, gflop / t_d if t_d else 0, t_w, gflop / t_w))
        tot['fwd'] += t_f * cnt
        tot['dgrad'] += t_d * cnt
        tot['wgrad'] += t_w * cnt
        totf += gflop * cnt
    print('total per step: fwd %.2f ms  dgrad %.2f ms  wgrad %.2f ms  sum %.2f ms   (%.1f GFLOP fwd -> %.0f TF/s overall)'
          % (tot['fwd'], tot['dgrad'], tot['wgrad'], sum(tot.values()), totf, 3 * totf / sum(tot.values())))


if __name__ == '__main__':
    main()

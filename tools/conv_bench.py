#!/usr/bin/env python3
"""Per-shape timing of the conv kernel (fwd / dgrad / wgrad) over the distinct ResNet-50 layer classes of
SURVEY.md Appendix A at batch 64.  Prints ms and TFLOP/s per class and the weighted total.  GPU box only."""
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

# Cin, Hin, Cout, k, stride, dil, count   (256x256 input, stride 16)
R50 = [(3, 256, 64, 7, 2, 1, 1), (64, 64, 64, 1, 1, 1, 1), (64, 64, 64, 3, 1, 1, 3), (64, 64, 256, 1, 1, 1, 4), (256, 64, 64, 1, 1, 1, 2),
       (256, 64, 128, 1, 1, 1, 1), (128, 64, 128, 3, 2, 1, 1), (128, 32, 512, 1, 1, 1, 4), (256, 64, 512, 1, 2, 1, 1),
       (512, 32, 128, 1, 1, 1, 3), (128, 32, 128, 3, 1, 1, 3), (512, 32, 256, 1, 1, 1, 1), (256, 32, 256, 3, 2, 1, 1),
       (256, 16, 1024, 1, 1, 1, 6), (512, 32, 1024, 1, 2, 1, 1), (1024, 16, 256, 1, 1, 1, 5), (256, 16, 256, 3, 1, 1, 5),
       (1024, 16, 512, 1, 1, 1, 1), (512, 16, 512, 3, 1, 2, 1), (512, 16, 2048, 1, 1, 1, 3), (1024, 16, 2048, 1, 1, 1, 1),
       (2048, 16, 512, 1, 1, 1, 2), (512, 16, 512, 3, 1, 1, 2), (2048, 16, 272, 3, 1, 1, 1)]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--only', default='')
    ap.add_argument('--mode', default='all', help='all | fwd | dgrad | wgrad')
    ap.add_argument('--img', action='store_true', help='also time the image-fed instances (pre-split activation / gradient images, pre-built weight images: what the block executor launches)')
    ap.add_argument('--tune', default='', help='p3d_fx_tune settings for this run, e.g. "7=1,8=0" (7: conv block order, 8: wgrad block order; -1 / unset = built-in choice)')
    a = ap.parse_args()
    for kv in filter(None, a.tune.split(',')):
        what, value = kv.split('=')
        L.p3d_fx_tune(int(what), int(value))
    tot = dict(fwd=0.0, dgrad=0.0, wgrad=0.0)
    tot_img = dict(fwd=0.0, dgrad=0.0, wgrad=0.0)
    totf = 0.0
    print('%-34s %8s | %8s %6s | %8s %6s | %8s %6s' % ('shape', 'GFLOP', 'fwd ms', 'TF', 'dgrad ms', 'TF', 'wgrad ms', 'TF'))
    for (c, h, k, ks, st, dil, cnt) in R50:
        tag = 'c%d h%d k%d %dx%d s%d d%d x%d' % (c, h, k, ks, ks, st, dil, cnt)
        if a.only and a.only not in tag:
            continue
        pad = dil * (ks - 1) // 2
        x = torch.randn(a.batch, c, h, h, device='cuda')
        w = torch.randn(k, c, ks, ks, device='cuda') * 0.05
        d = ops._desc(x.shape, w.shape, st, pad, dil)
        y = torch.empty(a.batch, k, d.Ho, d.Wo, device='cuda')
        dy = torch.randn_like(y)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        ws = ops.workspace(x.device, L.p3d_conv2d_wgrad_workspace_bytes(ctypes.byref(d)))
        st_ = ops._stream()
        p = ops._p
        gflop = 2.0 * a.batch * k * d.Ho * d.Wo * c * ks * ks / 1e9
        wsf = torch.empty(max(L.p3d_conv2d_fwd_workspace_bytes(ctypes.byref(d)), 16), dtype=torch.uint8, device='cuda')
        t_f = timeit(lambda: L.p3d_conv2d_fwd(ctypes.byref(d), p(x), p(w), None, None, None, p(y), p(wsf), wsf.numel(), st_), a.iters) if a.mode in ('all', 'fwd') else 1e9
        wsd = torch.empty(max(L.p3d_conv2d_dgrad_workspace_bytes(ctypes.byref(d)), 16), dtype=torch.uint8, device='cuda')
        t_d = timeit(lambda: L.p3d_conv2d_dgrad(ctypes.byref(d), p(dy), p(w), None, None, p(dx), p(wsd), wsd.numel(), st_), a.iters) if (c > 4 and a.mode in ('all', 'dgrad')) else 0.0
        t_w = timeit(lambda: L.p3d_conv2d_wgrad(ctypes.byref(d), p(dy), p(x), None, None, p(dw), p(ws), ws.numel(), st_), a.iters) if a.mode in ('all', 'wgrad') else 1e9
        print('%-34s %8.1f | %8.3f %6.1f | %8.3f %6.1f | %8.3f %6.1f' % (tag, gflop, t_f, gflop / t_f, t_d, gflop / t_d if t_d else 0, t_w, gflop / t_w))
        if a.img and c <= 4 and ks == 7 and L.p3d_stem_supported(a.batch, c, h, h, k):
            ximg = torch.empty(L.p3d_stem_image_bytes(a.batch, h, h), dtype=torch.uint8, device='cuda')
            wimg = torch.empty(L.p3d_stem_weight_image_bytes(k), dtype=torch.uint8, device='cuda')
            wsx = torch.empty(max(L.p3d_stem_workspace_bytes(a.batch, h, h, k), k * 1024), dtype=torch.uint8, device='cuda')
            L.p3d_stem_weight_image(p(w), k, c, p(wimg), p(wsx), wsx.numel(), st_)
            t_i = timeit(lambda: L.p3d_stem_image(p(x), p(ximg), a.batch, c, h, h, st_), a.iters)
            i_f = timeit(lambda: L.p3d_stem_fwd(p(ximg), p(wimg), p(y), a.batch, c, h, h, k, st_), a.iters)
            i_w = timeit(lambda: L.p3d_stem_wgrad(p(dy), p(ximg), p(dw), a.batch, c, h, h, k, 0, p(wsx), wsx.numel(), st_), a.iters)
            print('%-34s %8s | %8.3f %6.1f | %8s %6s | %8.3f %6.1f   space-to-depth image of x %.3f ms' % ('  restated (x3)', '', i_f, gflop / i_f, '', '', i_w, gflop / i_w, t_i))
            tot_img['fwd'] += i_f * cnt
            tot_img['wgrad'] += i_w * cnt
        if a.img and c % 16 == 0 and k % 16 == 0:
            fb, bb = ctypes.c_size_t(), ctypes.c_size_t()
            L.p3d_fx_weight_image_bytes(k, c, ks * ks, ctypes.byref(fb), ctypes.byref(bb))
            wf, wb = torch.empty(fb.value, dtype=torch.uint8, device='cuda'), torch.empty(bb.value, dtype=torch.uint8, device='cuda')
            L.p3d_fx_weight_images(p(w), k, c, ks * ks, p(wf), p(wb), st_)
            xi, dyi = ops.act_image(x), ops.act_image(dy)
            w0 = torch.empty(max(L.p3d_fx_conv_img_workspace_bytes(ctypes.byref(d), 0), 16), dtype=torch.uint8, device='cuda')
            w1 = torch.empty(max(L.p3d_fx_conv_img_workspace_bytes(ctypes.byref(d), 1), 16), dtype=torch.uint8, device='cuda')
            w2 = torch.empty(max(L.p3d_fx_conv_img_workspace_bytes(ctypes.byref(d), 2), 16), dtype=torch.uint8, device='cuda')
            i_f = timeit(lambda: L.p3d_fx_conv_fwd_img(ctypes.byref(d), p(xi), p(w), p(wf), None, p(y), p(w0), w0.numel(), st_), a.iters)
            i_d = timeit(lambda: L.p3d_fx_conv_dgrad_img(ctypes.byref(d), p(dyi), p(w), p(wb), p(dx), p(w1), w1.numel(), st_), a.iters)
            i_w = timeit(lambda: L.p3d_fx_conv_wgrad_img(ctypes.byref(d), p(dyi), None, p(xi), p(dw), p(w2), w2.numel(), st_), a.iters)
            i_w2 = timeit(lambda: L.p3d_fx_conv_wgrad_img(ctypes.byref(d), p(dyi), p(x), None, p(dw), p(w2), w2.numel(), st_), a.iters)
            t_img = timeit(lambda: ops.act_image(dy), a.iters)
            print('%-34s %8s | %8.3f %6.1f | %8.3f %6.1f | %8.3f %6.1f   wgrad(fp32 x) %.3f %.1f   image pass of dy %.3f ms (%.0f GB/s)'
                  % ('  image-fed', '', i_f, gflop / i_f, i_d, gflop / i_d, i_w, gflop / i_w, i_w2, gflop / i_w2, t_img, dy.numel() * 10 / t_img / 1e6))
            for key, v in (('fwd', i_f), ('dgrad', i_d), ('wgrad', i_w)):
                tot_img[key] += v * cnt
        tot['fwd'] += t_f * cnt
        tot['dgrad'] += t_d * cnt
        tot['wgrad'] += t_w * cnt
        totf += gflop * cnt
    print('total per step: fwd %.2f ms  dgrad %.2f ms  wgrad %.2f ms  sum %.2f ms   (%.1f GFLOP fwd -> %.1f TF/s overall)'
          % (tot['fwd'], tot['dgrad'], tot['wgrad'], sum(tot.values()), totf, 3 * totf / sum(tot.values())))
    if a.img:
        print('image-fed (shapes with C %% 16 == 0): fwd %.2f ms  dgrad %.2f ms  wgrad %.2f ms' % (tot_img['fwd'], tot_img['dgrad'], tot_img['wgrad']))


if __name__ == '__main__':
    main()

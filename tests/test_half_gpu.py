"""fp16 (-half_acc) kernels through the C ABI against the numpy oracle evaluated on the SAME fp16-rounded operands
(float64 accumulation): what remains is the fp32 summation order and the final rounding of the result to fp16."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import np_ops as ref

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def r16(a):
    return np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)


def nhwc16(a, cpad=None):
    """NCHW float numpy -> NHWC fp16 device tensor (channels zero-padded to cpad)"""
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda().permute(0, 2, 3, 1)
    if cpad and cpad > t.shape[-1]:
        t = torch.nn.functional.pad(t, (0, cpad - t.shape[-1]))
    return t.contiguous().half()


def nchw32(t, c=None):
    a = t.float().permute(0, 3, 1, 2).contiguous().cpu().numpy()
    return a if c is None else a[:, :c]


HCONV_CASES = [
    # name,      N, C,  H,  W,  K,  k, s, p, d
    ('1x1',      2, 64, 16, 16, 128, 1, 1, 0, 1),
    ('1x1s2',    2, 48, 17, 15, 40, 1, 2, 0, 1),
    ('3x3',      2, 32, 20, 20, 64, 3, 1, 1, 1),
    ('3x3s2',    3, 24, 33, 31, 72, 3, 2, 1, 1),
    ('3x3d2',    2, 16, 16, 16, 272, 3, 1, 2, 2),
    ('stem',     2, 3, 64, 64, 64, 7, 2, 3, 1),
    ('stem1',    1, 1, 65, 63, 64, 7, 2, 3, 1),
    ('big',      4, 256, 16, 16, 256, 3, 1, 1, 1),
    ('wideN',    2, 16, 64, 64, 64, 3, 1, 1, 1),
    ('deepK',    1, 512, 8, 8, 136, 1, 1, 0, 1),
]


@pytest.mark.parametrize('case', HCONV_CASES, ids=[c[0] for c in HCONV_CASES])
def test_hconv_fwd_dgrad_wgrad(case, pkg):
    L = pkg._lib.lib()
    ops = pkg.ops
    name, n, c, h, w, k, ks, st, pad, dil = case
    rng = np.random.default_rng(abs(hash(name)) % 2 ** 31)
    x = r16(rng.standard_normal((n, c, h, w)))
    wt = r16(rng.standard_normal((k, c, ks, ks)) / np.sqrt(c * ks * ks))
    bias = rng.standard_normal(k).astype(np.float32)
    cpad = (c + 7) // 8 * 8
    y_ref = ref.conv2d_fwd(x, wt, bias, st, pad, dil)
    dy = r16(rng.standard_normal(y_ref.shape))
    d = ops._desc((n, cpad, h, w), (k, cpad, ks, ks), st, pad, dil)
    stream = ops._stream()
    p = ops._p
    xt, dyt = nhwc16(x, cpad), nhwc16(dy)
    wm = torch.from_numpy(wt).cuda()                                  # fp32 master [K][C][R][S]
    krsc = torch.empty(k, ks, ks, cpad, dtype=torch.float16, device='cuda')
    crsk = torch.empty(cpad, ks, ks, k, dtype=torch.float16, device='cuda')
    pkg._lib.check(L.p3d_weight_images_f16(p(wm), p(krsc), p(crsk), k, c, ks * ks, cpad, stream), 'images')
    assert np.array_equal(krsc.float().cpu().numpy()[..., :c], wt.transpose(0, 2, 3, 1))
    assert np.array_equal(crsk.float().cpu().numpy()[:c], wt.transpose(1, 2, 3, 0))
    bt = torch.from_numpy(bias).cuda()
    y = torch.empty(n, d.Ho, d.Wo, k, dtype=torch.float16, device='cuda')
    pkg._lib.check(L.p3d_hconv2d_fwd(ctypes.byref(d), p(xt), p(krsc), p(bt), None, None, p(y), stream), 'fwd')
    assert relerr(nchw32(y), y_ref) < 1.5e-3
    # dgrad
    dx_ref = ref.conv2d_dgrad(dy, wt, x.shape, st, pad, dil)
    dx = torch.full((n, h, w, cpad), float('nan'), dtype=torch.float16, device='cuda')
    pkg._lib.check(L.p3d_hconv2d_dgrad(ctypes.byref(d), p(dyt), p(crsk), None, p(dx), stream), 'dgrad')
    got = nchw32(dx)
    assert np.isfinite(got).all()
    assert relerr(got[:, :c], dx_ref) < 1.5e-3
    assert not got[:, c:].any()
    # d->accumulate: dx += result (the gradient another consumer of the same input already wrote: GradJoin, the first conv of a block with an identity shortcut)
    base16 = r16(rng.standard_normal((n, c, h, w)))
    acc = nhwc16(base16, cpad)
    d.accumulate = 1
    pkg._lib.check(L.p3d_hconv2d_dgrad(ctypes.byref(d), p(dyt), p(crsk), None, p(acc), stream), 'dgrad accumulate')
    d.accumulate = 0
    assert relerr(nchw32(acc)[:, :c], base16 + dx_ref) < 2e-3
    # wgrad: fp32 result, accumulated onto a given master gradient with a scale
    dw_ref = ref.conv2d_wgrad(dy, x, wt.shape, st, pad, dil)
    base = rng.standard_normal(wt.shape).astype(np.float32)
    dw = torch.from_numpy(base.copy()).cuda()
    ws = torch.empty(max(L.p3d_hconv2d_wgrad_workspace_bytes(ctypes.byref(d)), 16), dtype=torch.uint8, device='cuda')
    d.accumulate = 1
    pkg._lib.check(L.p3d_hconv2d_wgrad(ctypes.byref(d), p(dyt), p(xt), None, p(dw), c, 0.5, p(ws), ws.numel(), stream), 'wgrad')
    assert relerr(dw.cpu().numpy() - base, 0.5 * dw_ref) < 2e-4
    d.accumulate = 0
    pkg._lib.check(L.p3d_hconv2d_wgrad(ctypes.byref(d), p(dyt), p(xt), None, p(dw), c, 1.0, p(ws), ws.numel(), stream), 'wgrad')
    assert relerr(dw.cpu().numpy(), dw_ref) < 2e-5


SUM_CASES = [('1x1', 2, 64, 16, 16, 128, 1, 1, 0, 1), ('1x1s2', 2, 48, 17, 15, 40, 1, 2, 0, 1), ('3x3', 2, 32, 20, 20, 64, 3, 1, 1, 1), ('3x3d2', 2, 16, 16, 16, 272, 3, 1, 2, 2),
             ('ragged', 3, 24, 13, 11, 72, 3, 1, 1, 1), ('big', 4, 256, 16, 16, 256, 3, 1, 1, 1),
             # BASELINE's batch: 2048 table rows (layer1: more rows than the 512 blocks of a stand-alone statistics pass) and the widest result
             ('full_layer1', 64, 64, 64, 64, 64, 3, 1, 1, 1), ('full_layer4', 64, 512, 16, 16, 2048, 1, 1, 0, 1)]


@pytest.mark.parametrize('case', SUM_CASES, ids=[c[0] for c in SUM_CASES])
def test_hconv_epilogue_sums(case, pkg):
    """p3d_hconv2d_fwd_stats / p3d_hconv2d_dgrad_sums: the tensor is bit-identical to the plain launch's, and the per-(pixel tile, channel) table sums to the
    BatchNorm statistics of the rounded output (forward) / to sum g, sum g * xhat of the ReLU-masked gradient (data gradient), the sums p3d_hbn_train_fwd / bwd take
    in passes of their own (depthnet.py:42-56,98-116: the BatchNorm behind / in front of every conv of a block)."""
    L = pkg._lib.lib()
    ops = pkg.ops
    name, n, c, h, w, k, ks, st, pad, dil = case
    rng = np.random.default_rng(abs(hash(name)) % 2 ** 31 + 1)
    x = r16(rng.standard_normal((n, c, h, w)))
    wt = r16(rng.standard_normal((k, c, ks, ks)) / np.sqrt(c * ks * ks))
    d = ops._desc((n, c, h, w), (k, c, ks, ks), st, pad, dil)
    stream, p = ops._stream(), ops._p
    xt = nhwc16(x)
    krsc = torch.empty(k, ks, ks, c, dtype=torch.float16, device='cuda')
    crsk = torch.empty(c, ks, ks, k, dtype=torch.float16, device='cuda')
    pkg._lib.check(L.p3d_weight_images_f16(p(torch.from_numpy(wt).cuda()), p(krsc), p(crsk), k, c, ks * ks, c, stream), 'images')
    y0 = torch.empty(n, d.Ho, d.Wo, k, dtype=torch.float16, device='cuda')
    y1 = torch.full_like(y0, float('nan'))
    pkg._lib.check(L.p3d_hconv2d_fwd(ctypes.byref(d), p(xt), p(krsc), None, None, None, p(y0), stream), 'fwd')
    rows = L.p3d_hconv2d_sum_rows(ctypes.byref(d), 0)
    assert rows == -(-n * d.Ho * d.Wo // 128)
    part = torch.full((rows, k // 8, 16), float('nan'), device='cuda')
    pkg._lib.check(L.p3d_hconv2d_fwd_stats(ctypes.byref(d), p(xt), p(krsc), p(y1), p(part), stream), 'fwd_stats')
    assert torch.equal(y0, y1)
    yf = y1.double().reshape(-1, k)
    tot = part.double().sum(0)                                   # [K/8][16]
    s1, s2 = tot[:, :8].reshape(-1), tot[:, 8:].reshape(-1)
    assert (s1 - yf.sum(0)).abs().max() <= 1e-5 * yf.abs().sum(0).max()
    assert (s2 - (yf * yf).sum(0)).abs().max() <= 1e-5 * (yf * yf).sum(0).max()
    if st != 1:
        return
    # data gradient: dx and the sums of the BatchNorm + ReLU layer whose output x was (its raw conv output c_prev, constants coef)
    dy = nhwc16(r16(rng.standard_normal((n, k, d.Ho, d.Wo))))
    c_prev = nhwc16(r16(rng.standard_normal((n, c, h, w))))
    coef = torch.from_numpy(np.stack([rng.uniform(0.5, 1.5, c), rng.standard_normal(c) * 0.3, rng.standard_normal(c) * 0.2, rng.uniform(0.5, 2.0, c)], 1).astype(np.float32)).cuda()
    dx0 = torch.empty(n, h, w, c, dtype=torch.float16, device='cuda')
    dx1 = torch.full_like(dx0, float('nan'))
    pkg._lib.check(L.p3d_hconv2d_dgrad(ctypes.byref(d), p(dy), p(crsk), None, p(dx0), stream), 'dgrad')
    rows = L.p3d_hconv2d_sum_rows(ctypes.byref(d), 1)
    part = torch.full((rows, c // 8, 16), float('nan'), device='cuda')
    pkg._lib.check(L.p3d_hconv2d_dgrad_sums(ctypes.byref(d), p(dy), p(crsk), p(dx1), p(c_prev), p(coef), p(part), stream), 'dgrad_sums')
    assert torch.equal(dx0, dx1)
    cf, gf = c_prev.float().reshape(-1, c), dx1.float().reshape(-1, c)
    live = torch.addcmul(coef[:, 1], cf, coef[:, 0]) > 0            # fmaf(x, sc, sh) > 0, in fp32 like the kernel
    g = torch.where(live, gf, torch.zeros_like(gf)).double()
    xhat = ((cf - coef[:, 2]) * coef[:, 3]).double()
    tot = part.double().sum(0)
    s1, s2 = tot[:, :8].reshape(-1), tot[:, 8:].reshape(-1)
    assert (s1 - g.sum(0)).abs().max() <= 1e-5 * g.abs().sum(0).max()
    assert (s2 - (g * xhat).sum(0)).abs().max() <= 1e-5 * (g * xhat).abs().sum(0).max()


@pytest.mark.parametrize('n,c,k,h', [(2, 64, 256, 16), (3, 32, 64, 9)])
def test_hbn_relu_mask_bytes(n, c, k, h, pkg):
    """The closing BatchNorm + shortcut + ReLU of a block (depthnet.py:52-56) with its statistics from the conv epilogue and the ReLU mask as one byte per 8 outputs:
    the bytes are [y > 0], and p3d_hbn_train_bwd_mask returns bit for bit what p3d_hbn_train_bwd returns from y."""
    L = pkg._lib.lib()
    ops = pkg.ops
    rng = np.random.default_rng(n + c + k)
    stream, p = ops._stream(), ops._p
    d = ops._desc((n, c, h, h), (k, c, 1, 1), 1, 0, 1)
    x = nhwc16(r16(rng.standard_normal((n, c, h, h))))
    krsc = nhwc16(r16(rng.standard_normal((k, c, 1, 1)) / np.sqrt(c)))
    res = nhwc16(r16(rng.standard_normal((n, k, h, h))))
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, k).astype(np.float32)).cuda()
    beta = torch.from_numpy((rng.standard_normal(k) * 0.2).astype(np.float32)).cuda()
    P = n * h * h
    y = torch.empty(n, h, h, k, dtype=torch.float16, device='cuda')
    out = torch.empty_like(y)
    rows = L.p3d_hconv2d_sum_rows(ctypes.byref(d), 0)
    part = torch.empty(rows, k // 8, 16, device='cuda')
    coef = torch.empty(k, 4, device='cuda')
    mask = torch.zeros(P * k // 8, dtype=torch.uint8, device='cuda')
    pkg._lib.check(L.p3d_hconv2d_fwd_stats(ctypes.byref(d), p(x), p(krsc), p(y), p(part), stream), 'fwd_stats')
    pkg._lib.check(L.p3d_hbn_train_fwd_partial(p(y), p(res), p(gamma), p(beta), None, None, p(out), p(coef), P, k, 0.1, 1e-5, 1, p(part), rows, p(mask), stream), 'fwd_partial')
    bits = (out.reshape(P, k // 8, 8) > 0).to(torch.int32) * (2 ** torch.arange(8, device='cuda', dtype=torch.int32))
    assert torch.equal(bits.sum(-1).to(torch.uint8).reshape(-1), mask)
    want = torch.relu(y.float() * coef[:, 0] + coef[:, 1] + res.float())
    assert (out.float() - want).abs().max() <= 2e-3 * want.abs().max()
    dy = nhwc16(r16(rng.standard_normal((n, k, h, h))))
    ws = torch.empty(L.p3d_hbn_workspace_bytes(k), dtype=torch.uint8, device='cuda')
    got = []
    for use_mask in (False, True):
        dx, dres = torch.empty_like(y), torch.empty_like(y)
        dg, db = torch.empty(k, device='cuda'), torch.empty(k, device='cuda')
        if use_mask:
            pkg._lib.check(L.p3d_hbn_train_bwd_mask(p(dy), p(y), p(mask), p(coef), p(dx), p(dres), p(dg), p(db), P, k, 0, p(ws), ws.numel(), stream), 'bwd_mask')
        else:
            pkg._lib.check(L.p3d_hbn_train_bwd(p(dy), p(y), p(out), p(coef), p(dx), p(dres), p(dg), p(db), P, k, 1, 0, p(ws), ws.numel(), stream), 'bwd')
        got.append((dx, dres, dg, db))
    for a, b in zip(*got):
        assert torch.equal(a, b)


@pytest.mark.parametrize('case', [(2, 16, 20, 20, 64, 3, 1, 1, 1), (2, 8, 33, 31, 72, 3, 2, 1, 1), (2, 64, 16, 16, 128, 1, 1, 0, 1), (1, 1, 65, 63, 64, 7, 2, 3, 1)])
def test_hconv_partial(case, pkg):
    """Partial conv on the fp16 kernels (mask in the operand fetch, mult in the epilogue, pre-scaled dy in backward) against the
    oracle's partial_conv_fwd / partial_conv_bwd on fp16-rounded operands."""
    L = pkg._lib.lib()
    ops = pkg.ops
    n, c, h, w, k, ks, st, pad, dil = case
    rng = np.random.default_rng(n * 1000 + c * 10 + ks)
    x = r16(rng.standard_normal((n, c, h, w)))
    mask = (rng.random((n, 1, h, w)) > 0.4).astype(np.float32)
    mask[0, 0, :6, :6] = 0                                             # an all-masked window
    wt = r16(rng.standard_normal((k, c, ks, ks)) / np.sqrt(c * ks * ks))
    cpad = (c + 7) // 8 * 8
    y_ref, mask_out, mult = ref.partial_conv_fwd(x, mask, wt, None, st, pad, dil)
    d = ops._desc((n, cpad, h, w), (k, cpad, ks, ks), st, pad, dil)
    stream, p = ops._stream(), ops._p
    xt = nhwc16(x, cpad)
    krsc = torch.empty(k, ks, ks, cpad, dtype=torch.float16, device='cuda')
    crsk = torch.empty(cpad, ks, ks, k, dtype=torch.float16, device='cuda')
    pkg._lib.check(L.p3d_weight_images_f16(p(torch.from_numpy(wt).cuda()), p(krsc), p(crsk), k, c, ks * ks, cpad, stream), 'images')
    mt, mu = torch.from_numpy(mask).cuda(), torch.from_numpy(np.ascontiguousarray(mult, dtype=np.float32)).cuda()
    y = torch.empty(n, d.Ho, d.Wo, k, dtype=torch.float16, device='cuda')
    pkg._lib.check(L.p3d_hconv2d_fwd(ctypes.byref(d), p(xt), p(krsc), None, p(mt), p(mu), p(y), stream), 'fwd')
    assert relerr(nchw32(y), y_ref) < 2e-3
    dy = r16(rng.standard_normal(y_ref.shape))
    dx_ref, dw_ref = ref.partial_conv_bwd(dy, x, mask, wt, mult, st, pad, dil)[:2]
    dyt = nhwc16(dy)
    scaled = torch.empty_like(dyt)
    pkg._lib.check(L.p3d_hscale_pixels(p(dyt), p(mu), p(scaled), n * d.Ho * d.Wo, k, stream), 'scale')
    if c >= 8:
        dx = torch.empty(n, h, w, cpad, dtype=torch.float16, device='cuda')
        pkg._lib.check(L.p3d_hconv2d_dgrad(ctypes.byref(d), p(scaled), p(crsk), p(mt), p(dx), stream), 'dgrad')
        assert relerr(nchw32(dx, c), dx_ref) < 3e-3
    dw = torch.zeros(k, c, ks, ks, device='cuda')
    ws = torch.empty(max(L.p3d_hconv2d_wgrad_workspace_bytes(ctypes.byref(d)), 16), dtype=torch.uint8, device='cuda')
    pkg._lib.check(L.p3d_hconv2d_wgrad(ctypes.byref(d), p(scaled), p(xt), p(mt), p(dw), c, 1.0, p(ws), ws.numel(), stream), 'wgrad')
    assert relerr(dw.cpu().numpy(), dw_ref) < 3e-3                     # dy * mult is rounded to fp16 once


def test_layout_converters(pkg):
    L = pkg._lib.lib()
    ops = pkg.ops
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 5, 7, 9)).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    out = torch.full((3, 7, 9, 8), float('nan'), dtype=torch.float16, device='cuda')
    pkg._lib.check(L.p3d_nchw_f32_to_nhwc_f16(ops._p(xt), ops._p(out), 3, 5, 63, 8, 2.0, ops._stream()), 'to nhwc')
    got = out.float().cpu().numpy()
    assert np.array_equal(got[..., :5], (2 * x).astype(np.float16).astype(np.float32).transpose(0, 2, 3, 1)) and not got[..., 5:].any()
    back = torch.empty(3, 8, 7, 9, dtype=torch.float32, device='cuda')
    pkg._lib.check(L.p3d_nhwc_f16_to_nchw_f32(ops._p(out), ops._p(back), 3, 8, 63, 0.5, ops._stream()), 'to nchw')
    assert np.array_equal(back.cpu().numpy()[:, :5], (2 * x).astype(np.float16).astype(np.float32) * 0.5)


@pytest.mark.parametrize('n,c,h,w,relu,with_res', [(4, 64, 16, 16, True, False), (3, 256, 9, 7, True, True), (2, 8, 33, 31, False, False),
                                                  (2, 2048, 4, 4, True, True), (5, 128, 8, 8, False, True), (64, 64, 32, 32, True, False)])
def test_hbn_train_fwd_bwd(n, c, h, w, relu, with_res, pkg):
    L = pkg._lib.lib()
    ops = pkg.ops
    p, st = ops._p, ops._stream()
    rng = np.random.default_rng(n * 1000 + c)
    x = r16(rng.standard_normal((n, c, h, w)) * 2 + 1)
    res = r16(rng.standard_normal((n, c, h, w))) if with_res else None
    gamma = (1 + 0.2 * rng.standard_normal(c)).astype(np.float32)
    beta = (0.3 * rng.standard_normal(c)).astype(np.float32)
    rm = rng.standard_normal(c).astype(np.float32)
    rv = (1 + rng.random(c)).astype(np.float32)
    y_ref, mean, invstd, nrm, nrv = ref.bn_train_fwd(x, gamma, beta, rm, rv)
    pre = y_ref + (res if with_res else 0)
    out_ref = np.maximum(pre, 0) if relu else pre
    P = n * h * w
    xt = nhwc16(x)
    rt = nhwc16(res) if with_res else None
    gt, bt, rmt, rvt = (torch.from_numpy(a.copy()).cuda() for a in (gamma, beta, rm, rv))
    yt = torch.empty_like(xt)
    coef = torch.empty(c, 4, device='cuda')
    ws = torch.empty(L.p3d_hbn_workspace_bytes(c), dtype=torch.uint8, device='cuda')
    pkg._lib.check(L.p3d_hbn_train_fwd(p(xt), p(rt), p(gt), p(bt), p(rmt), p(rvt), p(yt), p(coef), P, c, 0.1, 1e-5, int(relu),
                                       p(ws), ws.numel(), st), 'hbn fwd')
    assert np.abs(nchw32(yt) - out_ref).max() < 2e-3 * max(1.0, np.abs(out_ref).max())
    cf = coef.cpu().numpy()
    assert relerr(cf[:, 2], mean) < 1e-5 and relerr(cf[:, 3], invstd) < 1e-5 and relerr(cf[:, 0], invstd * gamma) < 1e-5
    assert relerr(rmt.cpu().numpy(), nrm) < 1e-5 and relerr(rvt.cpu().numpy(), nrv) < 1e-5
    # backward: the mask comes from the kernel's own fp16 output (or is recomputed from x when there is no residual)
    dy = r16(rng.standard_normal(x.shape))
    y_dev = nchw32(yt)
    g = (dy * (y_dev > 0)).astype(np.float32) if relu else dy
    dx_ref, dg_ref, db_ref = ref.bn_train_bwd(g, x, mean, invstd, gamma)
    dyt = nhwc16(dy)
    dxt = torch.empty_like(xt)
    drt = torch.empty_like(xt) if with_res else None
    dg, db = torch.ones(c, device='cuda'), torch.ones(c, device='cuda')
    y_arg = yt if (relu and with_res) else None
    pkg._lib.check(L.p3d_hbn_train_bwd(p(dyt), p(xt), p(y_arg), p(coef), p(dxt), p(drt), p(dg), p(db), P, c, int(relu), 1,
                                       p(ws), ws.numel(), st), 'hbn bwd')
    assert np.abs(nchw32(dxt) - dx_ref).max() < 3e-3 * max(1.0, np.abs(dx_ref).max())
    assert relerr(dg.cpu().numpy() - 1, dg_ref) < 2e-3 and relerr(db.cpu().numpy() - 1, db_ref) < 2e-3
    if with_res:
        assert np.array_equal(nchw32(drt), g)


def test_hbn_eval_fwd(pkg):
    L = pkg._lib.lib()
    ops = pkg.ops
    p, st = ops._p, ops._stream()
    rng = np.random.default_rng(3)
    n, c, h, w = 3, 32, 9, 9
    x = r16(rng.standard_normal((n, c, h, w)))
    gamma, beta = rng.standard_normal(c).astype(np.float32), rng.standard_normal(c).astype(np.float32)
    rm, rv = rng.standard_normal(c).astype(np.float32), (0.5 + rng.random(c)).astype(np.float32)
    y_ref = np.maximum(ref.bn_eval_fwd(x, gamma, beta, rm, rv), 0)
    xt = nhwc16(x)
    yt = torch.empty_like(xt)
    ws = torch.empty(L.p3d_hbn_workspace_bytes(c), dtype=torch.uint8, device='cuda')
    args = [torch.from_numpy(a).cuda() for a in (gamma, beta, rm, rv)]
    pkg._lib.check(L.p3d_hbn_eval_fwd(p(xt), None, p(args[0]), p(args[1]), p(args[2]), p(args[3]), p(yt), n * h * w, c, 1e-5, 1, p(ws), ws.numel(), st), 'eval')
    assert np.abs(nchw32(yt) - y_ref).max() < 2e-3 * max(1.0, np.abs(y_ref).max())


@pytest.mark.parametrize('shape', [(2, 8, 16, 16), (1, 16, 17, 15), (3, 64, 7, 9), (2, 64, 128, 128)])
def test_hmaxpool(shape, pkg):
    L = pkg._lib.lib()
    ops = pkg.ops
    p, st = ops._p, ops._stream()
    n, c, h, w = shape
    rng = np.random.default_rng(sum(shape))
    x = r16(np.maximum(rng.standard_normal(shape), 0))
    y_ref, idx_ref = ref.maxpool3x3s2_fwd(x)
    ho, wo = y_ref.shape[2:]
    xt = nhwc16(x)
    yt = torch.empty(n, ho, wo, c, dtype=torch.float16, device='cuda')
    it = torch.empty(n, ho, wo, c, dtype=torch.uint8, device='cuda')
    pkg._lib.check(L.p3d_hmaxpool3x3s2_fwd(p(xt), p(yt), p(it), n, h, w, c, st), 'pool')
    assert np.array_equal(nchw32(yt), y_ref)
    assert np.array_equal(it.permute(0, 3, 1, 2).cpu().numpy(), idx_ref)
    dy = r16(rng.standard_normal(y_ref.shape))
    dxt = torch.empty_like(xt)
    pkg._lib.check(L.p3d_hmaxpool3x3s2_bwd(p(nhwc16(dy)), p(it), p(dxt), n, h, w, c, st), 'pool bwd')
    want = ref.maxpool3x3s2_bwd(dy, idx_ref, x.shape)
    assert np.abs(nchw32(dxt) - want).max() < 2e-3 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize('case', ['half_r18_b2', 'half_fusion_r18_b2', 'half_partial_r18_b2', 'half_pfusion_r18_b2'])
def test_half_train_step_matches_reference_half(case, pkg):
    """-half_acc iterations against what the reference's own fp16 path (model.half() on torch's CPU half kernels, fp32
    copy_params, loss scale 32; depth_train.py:73-83,413-449) produced for the same weights and batches.  The two fp16
    implementations round at different places, so the bars are the fp16 noise measured between the reference's own fp32 and
    fp16 runs (loss 3e-5, joints 4e-4, clip norm 6e-4, per-tensor gradient norms: median 2e-3, worst 5e-2)."""
    import json
    from conftest import golden_path
    from test_step_gpu import build
    g = np.load(golden_path('step_%s.npz' % case))
    meta = json.loads(str(g['meta']))
    assert '-half_acc' in meta['extra']
    args, model, trainer = build(pkg, meta)
    assert trainer.half_acc and model._p3d_half and all(p.dtype == torch.float32 for p in model.parameters())
    model.train()
    trainer.adapt_learn_rate(1)
    for it in range(meta['iters']):
        c, d, tc, tv = pkg.synth.make_batch(meta['batch'], side=meta['side'], rank=0, step=it, invalid_frac=meta['invalid_frac'])
        depth = torch.from_numpy(d).cuda() if ('-do_fusion' in meta['extra'] or '-depth_only' in meta['extra']) else None
        loss = float(trainer.train_step(torch.from_numpy(c).cuda(), depth, torch.from_numpy(tc).cuda(), torch.from_numpy(tv).cuda()))
        assert abs(loss - g['losses'][it]) < 1e-3 * abs(g['losses'][it]), (it, loss, g['losses'][it])
        spec_sel = trainer.last_spec_cam.cpu().numpy().reshape(-1, 3)[tv.reshape(-1)]
        ref_sel = g['spec_sel_%d' % it]
        assert np.abs(spec_sel - ref_sel).max() < 1e-3 * np.abs(ref_sel).max()
        total = trainer.optimizer.total_norm(1.0 / args.grad_scaling)
        assert abs(total - g['clip_total'][it]) < 5e-3 * g['clip_total'][it], (total, g['clip_total'][it])
    assert trainer.skipped_steps == 0
    names = meta['names']
    grads = {n: p.grad.detach().cpu().numpy() / args.grad_scaling for n, p in zip(trainer.list_names, trainer.list_params)}
    gn = np.array([np.linalg.norm(grads[n].astype(np.float64)) for n in names])
    rel = np.abs(gn - g['grad_norms']) / np.maximum(g['grad_norms'], 1e-4 * g['grad_norms'].max())
    assert np.median(rel) < 1e-2 and rel.max() < 0.1, (np.median(rel), rel.max())
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    pn = np.array([np.linalg.norm(sd[n].astype(np.float64)) for n in names])
    assert np.abs(pn - g['param_norms']).max() < 1e-3 * g['param_norms'].max()           # the reference's state_dict is fp16
    # the fp16 weight images follow the masters after the step
    img = model.layer1[0].conv1._h_images
    want = model.layer1[0].conv1.weight.detach().half().permute(0, 2, 3, 1)
    assert torch.equal(img.krsc, want.contiguous())


def test_half_contract_batch_step_against_the_fp32_oracle(pkg):
    """The -half_acc step at BASELINE's workload (ResNet-50, 256 x 256, batch 64: depth_train.py:73-83,413-449 on the contract configuration) against the oracle's
    fp32 CPU port from the same deterministic weights and batch.  The reference's goldens for fp16 stop at batch 2; at batch 64 the fp16 path picks the plans the
    bench runs (2048-row sum tables from the conv epilogues, 384-block weight-gradient slab plans, the narrow 64-channel layout).  Two bars: against fp32, the
    noise of fp16 storage through 53 layers (loss 5e-3, joints 2e-2 of the largest coordinate, clip norm 5e-2); and, isolating this round's epilogue sums, the same
    step with the BatchNorm sums from stand-alone passes (both fp16, rounded at the same places: joints 3e-3)."""
    from oracle.torch_port import TorchPort
    from test_step_gpu import build
    meta = dict(model='resnet50', side=256, extra=['-stride', '16', '-depth', '16', '-depth_range', '1000', '-loss_div', '10', '-learn_rate', '5e-5',
                                                   '-weight_decay', '4e-5', '-grad_norm', '5', '-half_acc'])
    args, model, trainer = build(pkg, meta)
    assert trainer.half_acc
    model.train()
    trainer.adapt_learn_rate(1)
    lr = trainer.optimizer.param_groups[0]['lr']
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    port = TorchPort(pkg.synth.det_state_dict(shapes, 0), family='depthnet', model='resnet50')
    c, d, tc, tv = pkg.synth.make_batch(64, side=256, rank=0, step=0)
    want = port.train_step(c, d, tc, tv, lr=lr, weight_decay=4e-5, grad_norm=5.0, loss_div=10.0)
    loss = float(trainer.train_step(torch.from_numpy(c).cuda(), None, torch.from_numpy(tc).cuda(), torch.from_numpy(tv).cuda()))
    spec = trainer.last_spec_cam.cpu().numpy()
    assert trainer.skipped_steps == 0
    scale = np.abs(want['spec_cam']).max()
    assert abs(loss - want['loss']) < 5e-3 * abs(want['loss']), (loss, want['loss'])
    assert np.abs(spec - want['spec_cam']).max() < 2e-2 * scale
    total = trainer.optimizer.total_norm(1.0 / args.grad_scaling)
    assert abs(total - want['clip_total']) < 5e-2 * want['clip_total'], (total, want['clip_total'])
    # the same step with the sums from stand-alone passes
    L = pkg._lib.lib()
    before = L.p3d_hblock_fuse_sums(0)
    try:
        args2, model2, trainer2 = build(pkg, meta)
        model2.train()
        trainer2.adapt_learn_rate(1)
        loss2 = float(trainer2.train_step(torch.from_numpy(c).cuda(), None, torch.from_numpy(tc).cuda(), torch.from_numpy(tv).cuda()))
        spec2 = trainer2.last_spec_cam.cpu().numpy()
    finally:
        L.p3d_hblock_fuse_sums(before)
    assert abs(loss - loss2) < 2e-3 * abs(loss2), (loss, loss2)
    assert np.abs(spec - spec2).max() < 3e-3 * scale


def test_half_overflow_skips_the_step(pkg):
    import json
    from conftest import golden_path
    from test_step_gpu import build
    meta = json.loads(str(np.load(golden_path('step_half_r18_b2.npz'))['meta']))
    meta = dict(meta, side=128)
    args, model, trainer = build(pkg, meta)
    trainer.grad_scaling = 1e30                      # every fp16 gradient overflows
    model.train()
    trainer.adapt_learn_rate(1)
    before = model.conv1.weight.detach().clone()
    c, d, tc, tv = pkg.synth.make_batch(2, side=128, rank=0, step=0)
    trainer.train_step(torch.from_numpy(c).cuda(), None, torch.from_numpy(tc).cuda(), torch.from_numpy(tv).cuda())
    assert trainer.skipped_steps == 1 and trainer.optimizer.steps_taken() == 0
    assert torch.equal(before, model.conv1.weight.detach())


def test_half_eval_forward(pkg):
    """Trainer.test under -half_acc: frozen-BN fp16 forward; the record stays close to the fp32 one."""
    import json
    from conftest import golden_path
    from test_step_gpu import build
    meta = json.loads(str(np.load(golden_path('step_half_r18_b2.npz'))['meta']))
    args, model, trainer = build(pkg, meta)
    args32, model32, trainer32 = build(pkg, dict(meta, extra=[]))
    c, d, tc, tv = pkg.synth.make_batch(2, side=256, rank=1, step=0)
    x = torch.from_numpy(c).cuda()
    model.eval(); model32.eval()
    with torch.no_grad():
        z16, _ = model(x)
        z32, _ = model32(x)
    assert z16.dtype == torch.float32 and z16.shape == z32.shape
    assert (z16 - z32).abs().max() < 2e-2 * z32.abs().max()


def test_half_distillation_step_matches_reference_half(pkg):
    """-do_teach under -half_acc: fp16 teacher (fusionnet) and student (depthnet), fp32 feature-distillation loss on the converted
    feature maps, loss-scaled backward; against one distill_train iteration of the reference's own fp16 path."""
    import json
    from conftest import golden_path
    g = np.load(golden_path('distill_half.npz'))
    args = pkg.opts.parse(['-model', 'resnet18', '-suffix', 't', '-data_name', 'h36m', '-save_path', '/tmp/p3d', '-criterion', 'SmoothL1',
                           '-num_joints', '17', '-side_in', '128', '-do_teach', '-do_fusion', '-half_acc'])
    student = pkg.depthnet.resnet18(args, False)
    teacher = pkg.fusionnet.resnet18(args, False)
    for net, seed in ((student, 0), (teacher, 1)):
        det = pkg.synth.det_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed)
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in det.items()})
    trainer = pkg.depth_train.Trainer(args, student.cuda(), pkg.utils.get_info())
    trainer.set_teacher(teacher.cuda())
    assert teacher._p3d_half and student._p3d_half
    trainer.verbose = False
    c, d, tc, tv = pkg.synth.make_batch(2, side=128, rank=11, step=0)
    record = trainer.train(1, [tuple(torch.from_numpy(x) for x in (c, d, tc, tv, g['att']))])
    want = json.loads(str(g['record']))
    assert record['cam_train_loss'] == pytest.approx(want['cam_train_loss'], rel=2e-3)
    assert record['dist_train_loss'] == pytest.approx(want['dist_train_loss'], rel=5e-3)
    assert trainer.skipped_steps == 0 and trainer.optimizer.steps_taken() == 1
    names = json.loads(str(g['names']))
    sd = {k: v.detach().cpu().numpy() for k, v in student.state_dict().items()}
    pn = np.array([np.linalg.norm(sd[n].astype(np.float64)) for n in names])
    assert np.abs(pn - g['param_norms']).max() < 1e-3 * g['param_norms'].max()


@pytest.mark.parametrize('shape', [(64, 64, 64, 3, 1, 1), (256, 64, 512, 1, 2, 1), (128, 64, 128, 3, 2, 1), (1024, 16, 256, 1, 1, 1), (512, 16, 512, 3, 1, 2),
                                   (2048, 16, 272, 3, 1, 1)], ids=lambda s: 'c%d_h%d_k%d_%dx%d_s%d_d%d' % (s[0], s[1], s[2], s[3], s[3], s[4], s[5]))
def test_hconv_adjoint_identities_at_full_size(shape, pkg):
    """The fp16 kernels at batch 64: <dy, conv(x, w)> = <dgrad(dy), x> = <wgrad(dy, x), w> up to the fp16 rounding of the stored results."""
    from_half = pkg.ops_half
    c, h, k, ks, st, dil = shape
    pad = dil * (ks - 1) // 2
    conv = pkg.nn.Conv2d(c, k, ks, stride=st, padding=pad, dilation=dil, bias=False).cuda()
    gen = torch.Generator(device='cuda').manual_seed(c + k)
    with torch.no_grad():
        conv.weight.copy_((torch.randn(conv.weight.shape, device='cuda', generator=gen) / (c * ks * ks) ** 0.5).half().float())
    from_half.refresh_weights(conv)
    x = torch.randn(64, c, h, h, device='cuda', generator=gen).half().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = conv(x)
    dy = torch.randn(y.shape, device='cuda', generator=gen).half().contiguous(memory_format=torch.channels_last)
    y.backward(dy)
    torch.cuda.synchronize()
    form = (dy.double() * y.detach().double()).sum().item()
    via_w = (conv.weight.grad.double() * conv.weight.detach().double()).sum().item()
    via_x = (x.grad.double() * x.detach().double()).sum().item()
    scale = (dy.double().norm() * y.detach().double().norm()).item()
    assert abs(form - via_w) < 2e-3 * scale and abs(form - via_x) < 2e-3 * scale, (form, via_w, via_x)


def test_hrelu_and_skip_relu_network(pkg):
    ops = pkg.ops
    x = torch.randn(2, 16, 5, 7, device='cuda').half().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = ops.relu(x)
    dy = torch.randn_like(y)
    y.backward(dy)
    assert torch.equal(y.detach(), torch.clamp(x.detach(), min=0)) and torch.equal(x.grad, dy * (x.detach() > 0))
    # -skip_relu network under -half_acc: one training step runs and matches the fp32 loss closely
    flags = ['-model', 'resnet18', '-suffix', 't', '-data_name', 'h36m', '-save_path', '/tmp/p3d', '-criterion', 'SmoothL1', '-num_joints', '17',
             '-side_in', '128', '-skip_relu']
    losses = []
    for extra in ([], ['-half_acc']):
        args = pkg.opts.parse(flags + extra)
        torch.manual_seed(3)
        model = pkg.depth_main.create_model(args)[0].cuda().train()
        trainer = pkg.depth_train.Trainer(args, model, pkg.utils.get_info())
        trainer.verbose = False
        trainer.adapt_learn_rate(1)
        c, d, tc, tv = pkg.synth.make_batch(4, side=128, rank=2, step=0)
        losses.append(float(trainer.train_step(torch.from_numpy(c).cuda(), None, torch.from_numpy(tc).cuda(), torch.from_numpy(tv).cuda())))
    assert losses[1] == pytest.approx(losses[0], rel=2e-3)


def test_hbn_frozen_statistics_backward(pkg):
    """BatchNorm in eval mode inside a training step (-do_freeze, depthnet.py:158-161) on fp16: forward with the running statistics, backward
    dx = g * gamma / sqrt(var + eps), dgamma = sum g * xhat, dbeta = sum g; against torch autograd in fp32 on the same fp16-rounded inputs."""
    torch.manual_seed(5)
    for c, relu, with_res in ((64, True, True), (128, True, False), (256, False, False)):
        bn = pkg.nn.BatchNorm2d(c).cuda()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.3)
            bn.running_mean.normal_(0, 0.5)
            bn.running_var.uniform_(0.5, 2.0)
        bn.eval()
        x = torch.randn(4, c, 9, 7, device='cuda').half().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        res = torch.randn_like(x).requires_grad_(True) if with_res else None
        before = (bn.running_mean.clone(), bn.running_var.clone(), int(bn.num_batches_tracked))
        y = bn(x, res=res, relu=relu)
        dy = torch.randn_like(y)
        y.backward(dy)
        assert torch.equal(bn.running_mean, before[0]) and torch.equal(bn.running_var, before[1]) and int(bn.num_batches_tracked) == before[2]
        xf = x.detach().float().requires_grad_(True)
        rf = res.detach().float().requires_grad_(True) if with_res else None
        w, b = bn.weight.detach().clone().requires_grad_(True), bn.bias.detach().clone().requires_grad_(True)
        ref = torch.nn.functional.batch_norm(xf, bn.running_mean, bn.running_var, w, b, False, 0.1, bn.eps)
        if with_res:
            ref = ref + rf
        if relu:
            ref = torch.relu(ref)
        ref.backward(dy.float())
        assert (y.float() - ref).abs().max() < 2e-2
        keep = (ref.detach().abs() > 2e-2) | (not relu)                       # away from the ReLU switch (fp16 rounding of y can flip the mask)
        assert ((x.grad.float() - xf.grad).abs() * keep).max() < 2e-2
        assert (bn.weight.grad - w.grad).abs().max() < 2e-2 * w.grad.abs().max() + 0.05
        assert (bn.bias.grad - b.grad).abs().max() < 2e-2 * b.grad.abs().max() + 0.05
        if with_res:
            assert ((res.grad.float() - rf.grad).abs() * keep).max() < 1e-3


def test_half_frozen_distillation_step(pkg):
    """-do_teach -do_freeze under -half_acc: the distillation step runs with every BatchNorm of both networks frozen (running statistics stay
    put) and its losses agree with the fp32 path on the same weights and batch."""
    flags = ['-model', 'resnet18', '-suffix', 't', '-data_name', 'h36m', '-save_path', '/tmp/p3d', '-criterion', 'SmoothL1', '-num_joints', '17',
             '-side_in', '128', '-do_teach', '-do_fusion', '-do_freeze']
    records, stats = [], []
    for extra in ([], ['-half_acc']):
        args = pkg.opts.parse(flags + extra)
        student, teacher = pkg.depthnet.resnet18(args, False), pkg.fusionnet.resnet18(args, False)
        for net, seed in ((student, 0), (teacher, 1)):
            det = pkg.synth.det_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed)
            net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in det.items()})
        trainer = pkg.depth_train.Trainer(args, student.cuda(), pkg.utils.get_info())
        trainer.set_teacher(teacher.cuda())
        trainer.verbose = False
        mean0 = student.bn1.running_mean.clone()
        c, d, tc, tv = pkg.synth.make_batch(2, side=128, rank=11, step=0)
        att = np.ones((2, 1, 8, 8), np.float32)
        records.append(trainer.train(1, [tuple(torch.from_numpy(x) for x in (c, d, tc, tv, att))]))
        assert torch.equal(student.bn1.running_mean, mean0)
        stats.append(trainer.optimizer.steps_taken() if extra else 1)
    assert stats[1] == 1
    assert records[1]['cam_train_loss'] == pytest.approx(records[0]['cam_train_loss'], rel=5e-3)
    assert records[1]['dist_train_loss'] == pytest.approx(records[0]['dist_train_loss'], rel=2e-2)


@pytest.mark.parametrize('case', [('bottleneck', 256, 64, 1, 1, 4, 32, False), ('bottleneck', 256, 128, 2, 1, 4, 32, True), ('bottleneck', 512, 256, 1, 2, 2, 16, True),
                                  ('basic', 64, 64, 1, 1, 4, 32, False), ('basic', 64, 128, 2, 1, 4, 32, True),
                                  ('bottleneck', 256, 64, 1, 1, 64, 64, False)],          # a layer1 block at BASELINE's batch: 2048-row sum tables, the workspace sized for them
                         ids=lambda c: '%s_c%d_p%d_s%d_d%d%s' % (c[0], c[1], c[2], c[3], c[4], '_ds' if c[7] else ''))
def test_half_block_executor_equals_the_per_layer_path(case, pkg):
    """p3d_hblock_fwd / p3d_hblock_bwd (one C call per block and direction) run the same fp16 kernels in the same order as the per-layer autograd path
    (depthnet.py:40-56,96-116 under model.half()): with the BatchNorm sums from stand-alone passes (p3d_hblock_fuse_sums(0)) output, input gradient, every parameter
    gradient and the running statistics are bit-identical; with the sums from the conv epilogues (the default) the statistics are the same sums in another order, so the
    results agree to fp16 rounding (a value may land on the neighbouring fp16 number)."""
    import os
    import test_block_gpu as tb
    if os.environ.get('P3D_BLOCKS', '1') == '0':
        pytest.skip('the block executors are switched off in this run (P3D_BLOCKS=0)')
    kind, inplanes, planes, stride, dil, n, h, with_ds = case
    oh = pkg.ops_half
    block = tb.build(pkg, kind, inplanes, planes, stride, dil, with_ds, seed=5)
    oh.refresh_weights(block)
    gen = torch.Generator(device='cuda').manual_seed(3)
    x0 = torch.randn(n, inplanes, h, h, device='cuda', generator=gen).relu_().half().contiguous(memory_format=torch.channels_last)
    res = []
    L = pkg._lib.lib()
    for fused, sums in ((False, 0), (True, 0), (True, 1)):
        oh.HALF_BLOCKS = fused
        before = L.p3d_hblock_fuse_sums(sums)
        try:
            state = {k: v.clone() for k, v in block.state_dict().items()}
            block.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            assert oh.block_usable(block, x) == fused
            y = block(x)
            assert ('HResidualBlockFn' in type(y.grad_fn).__name__) == fused
            dy = torch.randn(y.shape, device='cuda', generator=torch.Generator(device='cuda').manual_seed(4)).half().contiguous(memory_format=torch.channels_last)
            y.backward(dy)
            pkg.ops.join_side_stream()
            torch.cuda.synchronize()
            res.append(dict(y=y.detach().clone(), dx=x.grad.clone(), grads={k: p.grad.clone() for k, p in block.named_parameters()},
                            buffers={k: v.clone() for k, v in block.state_dict().items() if 'running' in k}))
            block.load_state_dict(state)
        finally:
            oh.HALF_BLOCKS = True
            L.p3d_hblock_fuse_sums(before)
    a, b, c = res
    assert torch.equal(a['y'], b['y']) and torch.equal(a['dx'], b['dx'])
    for k in a['grads']:
        assert torch.equal(a['grads'][k], b['grads'][k]), k
    for k in a['buffers']:
        assert torch.equal(a['buffers'][k], b['buffers'][k]), k
    # sums from the epilogues: the same values up to fp16 rounding of intermediate tensors; in the backward pass a rounding step at a ReLU's zero crossing flips that
    # element's mask, so gradients are compared in the mean (an indexing or scaling mistake moves every element)
    close = lambda p, q, tol: (p.float() - q.float()).abs().max().item() <= tol * max(q.float().abs().max().item(), 1e-6)
    close_mean = lambda p, q, tol: (p.float() - q.float()).abs().mean().item() <= tol * max(q.float().abs().mean().item(), 1e-9)
    assert close(c['y'], a['y'], 4e-3) and close_mean(c['dx'], a['dx'], 1e-2)
    for k in a['grads']:
        assert close_mean(c['grads'][k], a['grads'][k], 2e-2), k
    for k in a['buffers']:
        assert close(c['buffers'][k], a['buffers'][k], 1e-4), k

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
    pkg._lib.check(L.p3d_hconv2d_fwd(ctypes.byref(d), p(xt), p(krsc), p(bt), p(y), stream), 'fwd')
    assert relerr(nchw32(y), y_ref) < 1.5e-3
    # dgrad
    dx_ref = ref.conv2d_dgrad(dy, wt, x.shape, st, pad, dil)
    dx = torch.full((n, h, w, cpad), float('nan'), dtype=torch.float16, device='cuda')
    pkg._lib.check(L.p3d_hconv2d_dgrad(ctypes.byref(d), p(dyt), p(crsk), p(dx), stream), 'dgrad')
    got = nchw32(dx)
    assert np.isfinite(got).all()
    assert relerr(got[:, :c], dx_ref) < 1.5e-3
    assert not got[:, c:].any()
    # wgrad: fp32 result, accumulated onto a given master gradient with a scale
    dw_ref = ref.conv2d_wgrad(dy, x, wt.shape, st, pad, dil)
    base = rng.standard_normal(wt.shape).astype(np.float32)
    dw = torch.from_numpy(base.copy()).cuda()
    ws = torch.empty(max(L.p3d_hconv2d_wgrad_workspace_bytes(ctypes.byref(d)), 16), dtype=torch.uint8, device='cuda')
    d.accumulate = 1
    pkg._lib.check(L.p3d_hconv2d_wgrad(ctypes.byref(d), p(dyt), p(xt), p(dw), c, 0.5, p(ws), ws.numel(), stream), 'wgrad')
    assert relerr(dw.cpu().numpy() - base, 0.5 * dw_ref) < 2e-4
    d.accumulate = 0
    pkg._lib.check(L.p3d_hconv2d_wgrad(ctypes.byref(d), p(dyt), p(xt), p(dw), c, 1.0, p(ws), ws.numel(), stream), 'wgrad')
    assert relerr(dw.cpu().numpy(), dw_ref) < 2e-5


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

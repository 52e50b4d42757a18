"""Per-kernel parity of the HIP path (through the C ABI) against the numpy oracle and the golden vectors
that came from the real reference.  Needs an MI355X: run with `-m gpu`."""
import json

import numpy as np
import pytest
import torch

from conftest import golden_path
from oracle import np_ops as ref

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


CONV_CASES = [
    # name,      N, C,  H,  W,  K, k, s, p, d, bias
    ('1x1',      2, 64, 16, 16, 128, 1, 1, 0, 1, False),
    ('1x1s2',    2, 48, 17, 15, 40, 1, 2, 0, 1, False),
    ('3x3',      2, 32, 20, 20, 64, 3, 1, 1, 1, False),
    ('3x3s2',    3, 24, 33, 31, 70, 3, 2, 1, 1, False),
    ('3x3d2',    2, 16, 16, 16, 272, 3, 1, 2, 2, False),
    ('3x3bias',  2, 40, 17, 17, 51, 3, 1, 1, 1, True),
    ('7x7s2',    2, 3, 64, 64, 64, 7, 2, 3, 1, False),
    ('7x7s2c1',  1, 1, 65, 63, 64, 7, 2, 3, 1, False),
    ('5x5',      1, 5, 12, 13, 9, 5, 1, 2, 1, True),
    ('big',      4, 256, 16, 16, 256, 3, 1, 1, 1, False),
    ('wideN',    2, 16, 64, 64, 64, 3, 1, 1, 1, False),
]


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_dgrad_wgrad(case, pkg):
    ops = pkg.ops
    name, n, c, h, w, k, ks, st, pad, dil, bias = case
    rng = np.random.default_rng(hash(name) % 2 ** 31)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = (rng.standard_normal((k, c, ks, ks)) / np.sqrt(c * ks * ks)).astype(np.float32)
    b = rng.standard_normal(k).astype(np.float32) if bias else None
    y_ref = ref.conv2d_fwd(x, wt, b, st, pad, dil)
    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
    xt = dev(x).requires_grad_(True)
    wtt = dev(wt).requires_grad_(True)
    bt = dev(b).requires_grad_(True) if bias else None
    y = ops.conv2d(xt, wtt, bt, st, pad, dil)
    y.backward(dev(dy))
    # fp32 MFMA is an exact fmaf chain: only the summation order differs from the float64 oracle
    assert relerr(host(y), y_ref) < 5e-6
    assert relerr(host(xt.grad), ref.conv2d_dgrad(dy, wt, x.shape, st, pad, dil)) < 5e-6
    assert relerr(host(wtt.grad), ref.conv2d_wgrad(dy, x, wt.shape, st, pad, dil)) < 2e-5
    if bias:
        assert relerr(host(bt.grad), ref.conv2d_bgrad(dy)) < 5e-6


def test_partial_conv_matches_reference_golden(pkg):
    """PartialConv module vs vectors produced by the reference's own class (partial_conv.py:32-57)."""
    g = np.load(golden_path('partial_conv.npz'))
    for m in json.loads(str(g['meta'])):
        n = m['name']
        wt = g[n + '.w']
        conv = pkg.partial_conv.PartialConv(wt.shape[1], wt.shape[0], kernel_size=m['k'], stride=m['stride'], padding=m['pad'],
                                            dilation=m['dil'], bias=m['bias']).cuda()
        with torch.no_grad():
            conv.weight.copy_(dev(wt))
            if m['bias']:
                conv.bias.copy_(dev(g[n + '.b']))
        x = dev(g[n + '.x']).requires_grad_(True)
        y, mo = conv(x, dev(g[n + '.mask']))
        assert np.array_equal(host(mo), g[n + '.mask_out']), n
        assert np.abs(host(y) - g[n + '.y']).max() < 1e-5 * max(np.abs(g[n + '.y']).max(), 1.0), n
        y.backward(dev(g[n + '.dy']))
        assert np.abs(host(x.grad) - g[n + '.dx']).max() < 1e-5 * max(np.abs(g[n + '.dx']).max(), 1.0), n
        assert np.abs(host(conv.weight.grad) - g[n + '.dw']).max() < 1e-5 * max(np.abs(g[n + '.dw']).max(), 1.0), n
        if m['bias']:       # the ((raw - b) * mult + b) * mask_out branch of partial_conv.py:48-51: d out / d b = mask_out
            assert np.abs(host(conv.bias.grad) - g[n + '.db']).max() < 1e-5 * max(np.abs(g[n + '.db']).max(), 1.0), n


def test_conv_cat_equals_conv_of_concat(pkg):
    ops = pkg.ops
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 24, 9, 9)).astype(np.float32)
    y = rng.standard_normal((2, 40, 9, 9)).astype(np.float32)
    wt = (rng.standard_normal((32, 64, 1, 1)) * 0.1).astype(np.float32)
    out_ref = ref.conv2d_fwd(np.concatenate([x, y], 1), wt)
    dy = rng.standard_normal(out_ref.shape).astype(np.float32)
    xt, yt, wtt = dev(x).requires_grad_(True), dev(y).requires_grad_(True), dev(wt).requires_grad_(True)
    out = ops.conv_cat1x1(xt, yt, wtt)
    out.backward(dev(dy))
    dcat = ref.conv2d_dgrad(dy, wt, (2, 64, 9, 9))
    assert relerr(host(out), out_ref) < 5e-6
    assert relerr(host(xt.grad), dcat[:, :24]) < 5e-6
    assert relerr(host(yt.grad), dcat[:, 24:]) < 5e-6
    assert relerr(host(wtt.grad), ref.conv2d_wgrad(dy, np.concatenate([x, y], 1), wt.shape)) < 2e-5


BN_CASES = [(4, 64, 16, 16, True, True), (3, 10, 17, 17, True, False), (2, 130, 8, 8, False, True), (5, 7, 5, 3, False, False),
            (64, 16, 32, 32, True, True),
            # the 16x16-stage shapes: a 64 KB channel slab with the recomputed mask, relu + residual, plain
            (64, 256, 16, 16, True, False), (8, 128, 16, 16, True, True), (3, 256, 4, 4, False, False)]


@pytest.mark.parametrize('n,c,h,w,relu,with_res', BN_CASES)
def test_bn_train_fwd_bwd(n, c, h, w, relu, with_res, pkg):
    ops = pkg.ops
    rng = np.random.default_rng(n * 1000 + c)
    x = (rng.standard_normal((n, c, h, w)) * 2 + 3).astype(np.float32)
    res = rng.standard_normal((n, c, h, w)).astype(np.float32) if with_res else None
    gamma = (1 + 0.2 * rng.standard_normal(c)).astype(np.float32)
    beta = (0.3 * rng.standard_normal(c)).astype(np.float32)
    rm = rng.standard_normal(c).astype(np.float32)
    rv = (1 + rng.random(c)).astype(np.float32)
    y_ref, mean, invstd, nrm, nrv = ref.bn_train_fwd(x, gamma, beta, rm, rv)
    pre = y_ref + (res if with_res else 0)
    out_ref = np.maximum(pre, 0) if relu else pre
    dy = rng.standard_normal(x.shape).astype(np.float32)
    g = dy * (out_ref > 0) if relu else dy
    dx_ref, dg_ref, db_ref = ref.bn_train_bwd(g.astype(np.float32), x, mean, invstd, gamma)

    xt, gt, bt = dev(x).requires_grad_(True), dev(gamma).requires_grad_(True), dev(beta).requires_grad_(True)
    rt = dev(res).requires_grad_(True) if with_res else None
    rmt, rvt = dev(rm), dev(rv)
    out = ops.batch_norm_act(xt, gt, bt, rmt, rvt, rt, relu, True, 0.1, 1e-5)
    out.backward(dev(dy))
    assert np.abs(host(out) - out_ref).max() < 2e-5
    assert relerr(host(rmt), nrm) < 1e-6 and relerr(host(rvt), nrv) < 1e-5
    assert relerr(host(xt.grad), dx_ref) < 5e-5
    assert relerr(host(gt.grad), dg_ref) < 2e-5 and relerr(host(bt.grad), db_ref) < 2e-5
    if with_res:
        assert np.array_equal(host(rt.grad), g.astype(np.float32))


def test_bn_eval_fwd_bwd(pkg):
    ops = pkg.ops
    rng = np.random.default_rng(11)
    n, c, h, w = 3, 20, 9, 9
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    gamma, beta = rng.standard_normal(c).astype(np.float32), rng.standard_normal(c).astype(np.float32)
    rm, rv = rng.standard_normal(c).astype(np.float32), (0.5 + rng.random(c)).astype(np.float32)
    y_ref = np.maximum(ref.bn_eval_fwd(x, gamma, beta, rm, rv), 0)
    dy = rng.standard_normal(x.shape).astype(np.float32)
    dx_ref, dg_ref, db_ref = ref.bn_eval_bwd((dy * (y_ref > 0)).astype(np.float32), x, gamma, beta, rm, rv)
    xt, gt, bt = dev(x).requires_grad_(True), dev(gamma).requires_grad_(True), dev(beta).requires_grad_(True)
    rmt, rvt = dev(rm), dev(rv)
    out = ops.batch_norm_act(xt, gt, bt, rmt, rvt, None, True, False, 0.1, 1e-5)
    out.backward(dev(dy))
    assert np.abs(host(out) - y_ref).max() < 1e-5
    assert np.array_equal(host(rmt), rm) and np.array_equal(host(rvt), rv)      # eval mode leaves the running stats alone
    assert relerr(host(xt.grad), dx_ref) < 1e-5
    assert relerr(host(gt.grad), dg_ref) < 1e-5 and relerr(host(bt.grad), db_ref) < 1e-5


@pytest.mark.parametrize('shape', [(2, 5, 16, 16), (1, 3, 17, 15), (3, 2, 7, 9), (2, 3, 10, 12), (1, 2, 9, 8), (2, 3, 5, 4), (2, 64, 128, 128)])
def test_maxpool_fwd_bwd_with_ties(shape, pkg):
    ops = pkg.ops
    rng = np.random.default_rng(sum(shape))
    x = np.maximum(rng.standard_normal(shape), 0).astype(np.float32)          # post-ReLU input: many exact ties at 0
    y_ref, idx = ref.maxpool3x3s2_fwd(x)
    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
    xt = dev(x).requires_grad_(True)
    y = ops.maxpool3x3s2(xt)
    y.backward(dev(dy))
    assert np.array_equal(host(y), y_ref)
    assert np.abs(host(xt.grad) - ref.maxpool3x3s2_bwd(dy, idx, x.shape)).max() < 1e-6


def test_relu(pkg):
    ops = pkg.ops
    x = np.random.default_rng(0).standard_normal((3, 4, 5, 7)).astype(np.float32)
    xt = dev(x).requires_grad_(True)
    y = ops.relu(xt)
    y.backward(torch.ones_like(y))
    assert np.array_equal(host(y), np.maximum(x, 0))
    assert np.array_equal(host(xt.grad), (x > 0).astype(np.float32))


def test_head_matches_reference_golden(pkg):
    """utils.to_heatmap + utils.decode vs the reference's own outputs and autograd gradients."""
    g = np.load(golden_path('head.npz'))
    for m in json.loads(str(g['meta'])):
        n = m['name']
        z = dev(g[n + '.z']).requires_grad_(True)
        heat = pkg.utils.to_heatmap(z, m['depth'], m['num_joints'], m['height'], m['width'])
        coords = pkg.utils.decode(heat, m['depth_range'])
        coords.backward(dev(g[n + '.dc']))
        assert relerr(host(coords), g[n + '.coords']) < 2e-6, n
        assert relerr(host(z.grad), g[n + '.dz']) < 5e-5, n


@pytest.mark.parametrize('criterion', ['SmoothL1', 'L1', 'MSE'])
@pytest.mark.parametrize('invalid', [0.0, 0.3])
def test_pose_loss(criterion, invalid, pkg):
    ops = pkg.ops
    rng = np.random.default_rng(5)
    b, j = 6, 17
    relat = (rng.standard_normal((b, j, 3)) * 8 + 1000).astype(np.float32)     # mixes |diff|<1 and >1 after /loss_div
    cam = (rng.standard_normal((b, j, 3)) * 8).astype(np.float32)
    val = rng.random((b, j)) >= invalid
    val[:, 16] = True
    loss_ref, spec_ref, drelat_ref = ref.pose_loss_fwd_bwd(relat, cam, val, 16, 10.0, criterion)
    rt = dev(relat).requires_grad_(True)
    loss, spec = ops.pose_loss(rt, dev(cam), torch.from_numpy(val).cuda(), 16, 10.0, criterion)
    loss.backward()
    assert abs(float(loss) - loss_ref) < 1e-5 * max(abs(loss_ref), 1)
    assert relerr(host(spec), spec_ref) < 1e-6
    assert relerr(host(rt.grad), drelat_ref) < 1e-5


def test_clip_and_adam_two_steps(pkg):
    """FlatAdam (l2norm + adam kernels) vs the oracle restatement of clip_grad_norm_ + optim.Adam, clip active and inactive."""
    rng = np.random.default_rng(9)
    shapes = [(7, 3, 3, 3), (7,), (5, 7, 1, 1), (13,)]
    ps = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    params = [torch.nn.Parameter(dev(p.copy())) for p in ps]
    opt = pkg.optim.FlatAdam([('p%d' % i, p) for i, p in enumerate(params)], lr=1e-3, weight_decay=4e-5)
    ms = [np.zeros_like(p) for p in ps]
    vs = [np.zeros_like(p) for p in ps]
    for step, gscale in ((1, 10.0), (2, 0.01)):          # first step clips (norm >> 5), second does not
        gs = [(rng.standard_normal(s) * gscale).astype(np.float32) for s in shapes]
        opt.zero_grad()
        for p, g in zip(params, gs):
            p.grad.copy_(dev(g))
        opt.clip_and_step(5.0)
        total, coef = ref.clip_grad_norm(gs, 5.0)
        assert abs(opt.total_norm() - total) < 1e-5 * total
        for i in range(len(ps)):
            ps[i], ms[i], vs[i] = ref.adam_step(ps[i], gs[i], ms[i], vs[i], step, 1e-3, weight_decay=4e-5, grad_scale=coef)
            assert np.abs(host(params[i]) - ps[i]).max() < 2e-6, (step, i)


def test_ops_refuse_cpu_tensors(pkg):
    with pytest.raises(pkg._lib.P3DError):
        pkg.ops.conv2d(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3))


def test_augment_colour_and_erase(pkg):
    """On-GPU colour augmentation and eraser (config 5) against the numpy restatement (HSV part: parity unpinned, see oracle)."""
    ops = pkg.ops
    rng = np.random.default_rng(21)
    b, h, w = 3, 32, 40
    img = np.floor(rng.random((b, 3, h, w)) * 256).astype(np.float32)
    img[0, :, :4, :4] = 128.0                                       # a grey patch: zero saturation, hue undefined
    img[1, :, 5, 5] = [255, 0, 0]
    params = np.stack([rng.uniform(-0.125, 0.125, b), rng.uniform(0.8, 1.25, b), rng.uniform(-18, 18, b), rng.uniform(0.8, 1.25, b)], 1).astype(np.float32)
    want = np.stack([ref.augment_colour(img[i].transpose(1, 2, 0), *params[i]).transpose(2, 0, 1) for i in range(b)])
    got = host(ops.augment_colour_(dev(img), dev(params)))
    diff = np.abs(got - want)
    assert (diff > 1).sum() == 0 and (diff > 0).mean() < 0.01      # at most a rounding flip of the final truncation
    rects = np.array([ref.erase_rect((h, w), 0.2, 1.5, (0.3, 0.6)), ref.erase_rect((h, w), 0.1, 0.5, (1.0, 1.0)), (5, 5, 5, 9)], dtype=np.int32)
    colour = np.floor(rng.random((b, 3)) * 256).astype(np.float32)
    want = img.copy()
    for i, (x0, y0, x1, y1) in enumerate(rects):
        want[i, :, max(y0, 0):y1, max(x0, 0):x1] = colour[i][:, None, None]
    got = host(ops.augment_erase_(dev(img), torch.from_numpy(rects).cuda(), dev(colour)))
    assert np.array_equal(got, want)


def test_normalize_rgb_and_gpu_augment_pipeline(pkg):
    """ToTensor + Normalize of the loader (depth_datasets.py:78-79,91-93) on the GPU, and the GpuAugment pipeline order."""
    ops = pkg.ops
    rng = np.random.default_rng(3)
    img = np.floor(rng.random((2, 3, 16, 20)) * 256).astype(np.float32)
    mean, std = np.array(ops.IMAGENET_MEAN, np.float32), np.array(ops.IMAGENET_STD, np.float32)
    want = (img / np.float32(255) - mean[None, :, None, None]) / std[None, :, None, None]
    got = host(ops.normalize_rgb_(dev(img)))
    assert np.abs(got - want).max() < 2e-6
    aug = pkg.augment.GpuAugment(colour=True, eraser=True, seed=5)
    out = host(aug(dev(img), train=True))
    # same draws, applied by hand: colour jitter -> erase -> normalise
    rs = np.random.Generator(np.random.PCG64(5))
    params = pkg.augment.draw_colour_params(2, rs)
    rects, colour = pkg.augment.draw_erase_rects(2, 16, 20, rs)
    step = ops.augment_colour_(dev(img), dev(params))
    step = ops.augment_erase_(step, torch.from_numpy(rects).cuda(), dev(colour))
    assert np.array_equal(out, host(ops.normalize_rgb_(step)))
    assert np.array_equal(host(aug(dev(img), train=False)), got)                 # evaluation: normalisation only


def test_fusion_training_with_gpu_augmentation(pkg):
    """BASELINE config 5 wiring: fusionnet + -colour -eraser on the synthetic loader (raw 0..255 colour crops -> GPU augmentation)."""
    args = pkg.opts.parse(['-model', 'resnet18', '-suffix', 't', '-data_name', 'h36m', '-save_path', '/tmp/p3d', '-criterion', 'SmoothL1',
                           '-num_joints', '17', '-side_in', '128', '-do_fusion', '-colour', '-eraser', '-synthetic', '2', '-batch_size', '2',
                           '-workers', '0'])
    loader = pkg.depth_datasets.data_loader(args, 'train', pkg.utils.get_info())
    color = next(iter(loader))[0]
    assert float(color.max()) > 200 and float(color.min()) >= 0                   # raw crops, not normalised ones
    model, _ = pkg.depth_main.create_model(args)
    trainer = pkg.depth_train.Trainer(args, model.cuda(), pkg.utils.get_info())
    trainer.verbose = False
    assert trainer.gpu_augment is not None
    rec = trainer.train(1, loader)
    assert np.isfinite(rec['cam_train_loss']) and rec['cam_train_loss'] > 0


@pytest.mark.parametrize('dtype,chan', [(np.uint8, 3), (np.float32, 1)])
def test_warp_crops(dtype, chan, pkg):
    """On-GPU crop re-projection against the numpy restatement of cameralib.reproject_image_fast's remap."""
    ops = pkg.ops
    rng = np.random.default_rng(8)
    b, hs, ws, side = 3, 60, 80, 48
    frames = (rng.random((b, hs, ws, chan)) * 255).astype(dtype)
    homs = []
    for i in range(b):
        f_old, f_new = 70.0 + 10 * i, 90.0 + 20 * i                     # a zoomed, rotated, re-centred virtual camera (no parallax)
        k_old = np.array([[f_old, 0, ws / 2], [0, f_old, hs / 2], [0, 0, 1]])
        k_new = np.array([[f_new, 0, side / 2], [0, f_new, side / 2], [0, 0, 1]])
        ang = 0.15 * (i - 1)
        r_y = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        r_x = np.array([[1, 0, 0], [0, np.cos(0.1 * i), -np.sin(0.1 * i)], [0, np.sin(0.1 * i), np.cos(0.1 * i)]])
        r_new = r_y @ r_x
        homs.append(ref.crop_homography(k_old, np.eye(3), k_new, r_new))
    homs = np.stack(homs)
    got = host(ops.warp_crops(torch.from_numpy(frames).cuda(), dev(homs), (side, side)))
    want = np.stack([ref.warp_crop(frames[i], homs[i], (side, side)) for i in range(b)])
    assert got.shape == (b, chan, side, side)
    diff = np.abs(got - want)
    if dtype == np.uint8:
        assert (diff > 1).sum() == 0 and (diff > 0).mean() < 0.01       # a rounding flip where the interpolant sits on .5
    else:
        assert diff.max() < 1e-3 * 255
    assert (want == 0).mean() > 0.02 and (want > 0).mean() > 0.5        # the case crosses the frame border and the interior


@pytest.mark.parametrize('case', [(2, 64, 16, 16, 128, 1, 1, 0, 1, True, True), (2, 32, 20, 20, 64, 3, 1, 1, 1, False, True),
                                  (3, 24, 33, 31, 72, 3, 2, 1, 1, False, False), (4, 256, 16, 16, 256, 3, 1, 1, 1, True, True),
                                  (2, 3, 64, 64, 64, 7, 2, 3, 1, False, True)])
def test_conv_bn_eval_fused(case, pkg):
    """Inference kernel conv + frozen BatchNorm (+ residual + ReLU) against the oracle's conv followed by bn_eval (incl. the split-K shape)."""
    n, c, h, w, k, ks, st, pad, dil, with_res, relu = case
    rng = np.random.default_rng(n * 100 + c)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = (rng.standard_normal((k, c, ks, ks)) / np.sqrt(c * ks * ks)).astype(np.float32)
    gamma, beta = (1 + 0.2 * rng.standard_normal(k)).astype(np.float32), (0.3 * rng.standard_normal(k)).astype(np.float32)
    rm, rv = (0.2 * rng.standard_normal(k)).astype(np.float32), (0.5 + rng.random(k)).astype(np.float32)
    y = ref.bn_eval_fwd(ref.conv2d_fwd(x, wt, None, st, pad, dil), gamma, beta, rm, rv)
    res = rng.standard_normal(y.shape).astype(np.float32) if with_res else None
    want = y + (res if with_res else 0)
    want = np.maximum(want, 0) if relu else want
    conv = pkg.nn.Conv2d(c, k, ks, stride=st, padding=pad, dilation=dil, bias=False).cuda()
    bn = pkg.nn.BatchNorm2d(k).cuda().eval()
    with torch.no_grad():
        conv.weight.copy_(dev(wt)); bn.weight.copy_(dev(gamma)); bn.bias.copy_(dev(beta)); bn.running_mean.copy_(dev(rm)); bn.running_var.copy_(dev(rv))
        assert pkg.ops.can_fuse_eval(dev(x), conv, bn)
        got = host(pkg.ops.conv_bn_eval(dev(x), conv, bn, res=None if res is None else dev(res), relu=relu))
    assert np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(want).max())
    assert not pkg.ops.can_fuse_eval(dev(x), conv, bn)                           # with autograd on, the unfused path runs


R50_CLASSES = [(3, 256, 64, 7, 2, 1), (64, 64, 64, 1, 1, 1), (64, 64, 64, 3, 1, 1), (64, 64, 256, 1, 1, 1), (256, 64, 64, 1, 1, 1), (256, 64, 128, 1, 1, 1),
               (128, 64, 128, 3, 2, 1), (128, 32, 512, 1, 1, 1), (256, 64, 512, 1, 2, 1), (512, 32, 128, 1, 1, 1), (128, 32, 128, 3, 1, 1),
               (512, 32, 256, 1, 1, 1), (256, 32, 256, 3, 2, 1), (256, 16, 1024, 1, 1, 1), (512, 32, 1024, 1, 2, 1), (1024, 16, 256, 1, 1, 1),
               (256, 16, 256, 3, 1, 1), (1024, 16, 512, 1, 1, 1), (512, 16, 512, 3, 1, 2), (512, 16, 2048, 1, 1, 1), (1024, 16, 2048, 1, 1, 1),
               (2048, 16, 512, 1, 1, 1), (512, 16, 512, 3, 1, 1), (2048, 16, 272, 3, 1, 1)]


@pytest.mark.parametrize('shape', R50_CLASSES, ids=['c%d_h%d_k%d_%dx%d_s%d_d%d' % (s[0], s[1], s[2], s[3], s[3], s[4], s[5]) for s in R50_CLASSES])
def test_conv_adjoint_identities_at_full_size(shape, pkg):
    """Size-independent parity property at BASELINE's full sizes (batch 64, every conv class of ResNet-50):
    forward, dgrad and wgrad are the three faces of one trilinear form, <dy, conv(x, w)> = <dgrad(dy, w), x> = <wgrad(dy, x), w>.
    This exercises the production plans (split-K forward / dgrad, parity-class dgrad, split wgrad + fold, tap-major weight image)."""
    ops = pkg.ops
    c, h, k, ks, st, dil = shape
    pad = dil * (ks - 1) // 2
    gen = torch.Generator(device='cuda').manual_seed(c * 7 + k)
    x = torch.randn(64, c, h, h, device='cuda', generator=gen).requires_grad_(c > 4)
    w = (torch.randn(k, c, ks, ks, device='cuda', generator=gen) / (c * ks * ks) ** 0.5).requires_grad_(True)
    y = ops.conv2d(x, w, None, st, pad, dil)
    dy = torch.randn(y.shape, device='cuda', generator=gen)
    y.backward(dy)
    torch.cuda.synchronize()
    form = (dy.double() * y.detach().double()).sum().item()
    via_w = (w.grad.double() * w.detach().double()).sum().item()
    scale = (dy.double().norm() * y.detach().double().norm()).item()
    assert abs(form - via_w) < 2e-5 * scale, (form, via_w)
    if c > 4:                                            # (the stem never computes an input gradient)
        via_x = (x.grad.double() * x.detach().double()).sum().item()
        assert abs(form - via_x) < 2e-5 * scale, (form, via_x)


@pytest.mark.parametrize('shape', R50_CLASSES[1:], ids=['c%d_h%d_k%d_%dx%d_s%d_d%d' % (s[0], s[1], s[2], s[3], s[3], s[4], s[5]) for s in R50_CLASSES[1:]])
def test_image_fed_adjoint_identities_at_full_size(shape, pkg):
    """The same trilinear-form property for the IMAGE-FED instances the block executor launches (p3d_fx_conv_*_img on pre-split activation / gradient images) at
    BASELINE's batch 64: the 64- and 96-row tiles of fx16_conv_kernel, the tap-inner K order of the regressor, the one-launch strided data gradient, the two- / three-tap
    column tiles of the 64-channel weight gradients and every split / slab plan of the production sizes; and each result against the fp32-fed kernels on the same data."""
    ops = pkg.ops
    c, h, k, ks, st, dil = shape
    pad = dil * (ks - 1) // 2
    gen = torch.Generator(device='cuda').manual_seed(c * 5 + k + ks)
    x = torch.randn(64, c, h, h, device='cuda', generator=gen)
    w = torch.randn(k, c, ks, ks, device='cuda', generator=gen) / (c * ks * ks) ** 0.5
    x_img = ops.act_image(x)
    y = ops.conv2d_img('fwd', x.shape, w, st, pad, dil, x_img=x_img)
    dy = torch.randn(y.shape, device='cuda', generator=gen)
    dy_img = ops.act_image(dy)
    dx = ops.conv2d_img('dgrad', x.shape, w, st, pad, dil, dy_img=dy_img)
    dw = ops.conv2d_img('wgrad', x.shape, w, st, pad, dil, dy_img=dy_img, x_img=x_img)
    torch.cuda.synchronize()
    form = (dy.double() * y.double()).sum().item()
    scale = (dy.double().norm() * y.double().norm()).item()
    assert abs(form - (dw.double() * w.double()).sum().item()) < 2e-5 * scale
    assert abs(form - (dx.double() * x.double()).sum().item()) < 2e-5 * scale
    # against the fp32-fed kernels (the in-kernel split): same products, another accumulation order
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y0 = ops.conv2d(xr, wr, None, st, pad, dil)
    y0.backward(dy)
    ops.join_side_stream()
    torch.cuda.synchronize()
    for name, got, ref_t in (('fwd', y, y0.detach()), ('dgrad', dx, xr.grad), ('wgrad', dw, wr.grad)):
        assert (got - ref_t).abs().max().item() <= 1e-5 * ref_t.abs().max().item(), name


# Sampled-oracle parity at BASELINE's batch: the plans the bench runs (tile choice, split-K, parity classes, slab folds, weight images,
# x3 kernels) are compared with the float64 oracle itself, evaluated only at randomly chosen outputs.  Beyond the ResNet-50 classes:
# the shapes only the fusion / partial networks have (fusionnet.py:130-140: 2C -> C 1x1 over the concat at 32x32, batch 32; the 1-channel
# 7x7 stride-2 depth stem) and the MASKED (partial_conv.py:32-57) variants of the layer1 / layer2 classes at batch 64.
FULL_CASES = ([('r50', 64) + s + (False,) for s in R50_CLASSES]
              + [('stem1', 32, 1, 256, 64, 7, 2, 1, False), ('stem1', 64, 1, 256, 64, 7, 2, 1, True)]
              + [('masked', 64) + s + (True,) for s in R50_CLASSES[1:8] + R50_CLASSES[9:11]])


def _sample_idx(rng, shape, count):
    return np.stack([rng.integers(0, d, count) for d in shape], 1)


@pytest.mark.parametrize('case', FULL_CASES, ids=['%s_n%d_c%d_h%d_k%d_%dx%d_s%d_d%d%s' % (c[0], c[1], c[2], c[3], c[4], c[5], c[5], c[6], c[7], '_masked' if c[8] else '')
                                                  for c in FULL_CASES])
def test_conv_sampled_oracle_at_full_size(case, pkg):
    ops = pkg.ops
    tag, n, c, h, k, ks, st, dil, masked = case
    pad = dil * (ks - 1) // 2
    gen = torch.Generator(device='cuda').manual_seed(c * 11 + k + ks)
    rng = np.random.default_rng(c * 13 + k)
    need_dx = c > 4
    x = torch.randn(n, c, h, h, device='cuda', generator=gen).requires_grad_(need_dx)
    w = (torch.randn(k, c, ks, ks, device='cuda', generator=gen) / (c * ks * ks) ** 0.5).requires_grad_(True)
    mask = mult = None
    if masked:
        mask = (torch.rand(n, 1, h, h, device='cuda', generator=gen) >= 0.3).float()
        mult, _ = ops.mask_count(mask, ks, st, pad, dil)
    y = ops.conv2d(x, w, None, st, pad, dil, mask_in=mask, mult=mult)
    dy = torch.randn(y.shape, device='cuda', generator=gen)
    y.backward(dy)
    torch.cuda.synchronize()
    xe = x.detach() * mask if masked else x.detach()              # the oracle sees the operands the reference would form (partial_conv.py:45-53)
    dye = dy * mult if masked else dy
    xh, wh, dyh = host(xe), host(w), host(dye)
    count = 4096
    # forward
    idx = _sample_idx(rng, y.shape, count)
    want = ref.conv2d_fwd_at(xh, wh, None, st, pad, dil, idx)
    if masked:
        want = want * host(mult)[idx[:, 0], 0, idx[:, 2], idx[:, 3]]
    got = host(y)[tuple(idx.T)]
    assert np.abs(got - want).max() < 2e-5 * np.abs(want).max(), 'fwd'
    # data gradient
    if need_dx:
        idx = _sample_idx(rng, x.shape, count)
        want = ref.conv2d_dgrad_at(dyh, wh, x.shape, st, pad, dil, idx)
        if masked:
            want = want * host(mask)[idx[:, 0], 0, idx[:, 2], idx[:, 3]]
        got = host(x.grad)[tuple(idx.T)]
        assert np.abs(got - want).max() < 2e-5 * np.abs(want).max(), 'dgrad'
    # weight gradient: a random block of filters x input channels, every tap (>= 4096 outputs)
    nk = min(k, 64)
    nc = min(c, max(4096 // (nk * ks * ks) + 1, min(c, 64)))
    ksel, csel = np.sort(rng.choice(k, nk, replace=False)), np.sort(rng.choice(c, nc, replace=False))
    want = ref.conv2d_wgrad_block(dyh, xh, ksel, csel, ks, ks, st, pad, dil)
    assert want.size >= min(4096, k * c * ks * ks)
    got = host(w.grad)[ksel][:, csel]
    assert np.abs(got - want).max() < 5e-5 * np.abs(want).max(), 'wgrad'


def test_conv_cat_sampled_oracle_at_fusion_size(pkg):
    """fusionnet.Fusion's 1x1 over cat([rgb, depth]) (fusionnet.py:138-139) at BASELINE config 5's size: 2 x 512 -> 512 channels at 32x32, batch 32."""
    ops = pkg.ops
    gen = torch.Generator(device='cuda').manual_seed(77)
    rng = np.random.default_rng(78)
    xa = torch.randn(32, 512, 32, 32, device='cuda', generator=gen).requires_grad_(True)
    xb = torch.randn(32, 512, 32, 32, device='cuda', generator=gen).requires_grad_(True)
    w = (torch.randn(512, 1024, 1, 1, device='cuda', generator=gen) / 32.0).requires_grad_(True)
    y = ops.conv_cat1x1(xa, xb, w)
    dy = torch.randn(y.shape, device='cuda', generator=gen)
    y.backward(dy)
    torch.cuda.synchronize()
    cat = np.concatenate([host(xa), host(xb)], axis=1)
    wh, dyh = host(w), host(dy)
    idx = _sample_idx(rng, y.shape, 4096)
    want = ref.conv2d_fwd_at(cat, wh, None, 1, 0, 1, idx)
    assert np.abs(host(y)[tuple(idx.T)] - want).max() < 2e-5 * np.abs(want).max()
    idx = _sample_idx(rng, cat.shape, 4096)
    want = ref.conv2d_dgrad_at(dyh, wh, cat.shape, 1, 0, 1, idx)
    got = np.concatenate([host(xa.grad), host(xb.grad)], axis=1)[tuple(idx.T)]
    assert np.abs(got - want).max() < 2e-5 * np.abs(want).max()
    ksel, csel = np.sort(rng.choice(512, 64, replace=False)), np.sort(rng.choice(1024, 64, replace=False))
    want = ref.conv2d_wgrad_block(dyh, cat, ksel, csel, 1, 1, 1, 0, 1)
    assert np.abs(host(w.grad)[ksel][:, csel] - want).max() < 5e-5 * np.abs(want).max()


@pytest.mark.parametrize('shape', [(64, 64, 128, 128), (64, 256, 64, 64), (64, 1024, 16, 16), (64, 2048, 16, 16)])
def test_bn_properties_at_full_size(shape, pkg):
    """Size-independent BatchNorm properties at the contract's tensor sizes: the training output has zero mean / unit variance per channel,
    and the input gradient is orthogonal to 1 and to the normalised input (what subtracting k1 + xhat * k2 means)."""
    ops = pkg.ops
    n, c, h, w = shape
    gen = torch.Generator(device='cuda').manual_seed(c)
    x = (torch.randn(shape, device='cuda', generator=gen) * 3 + 1.5).requires_grad_(True)
    gamma, beta = torch.ones(c, device='cuda', requires_grad=True), torch.zeros(c, device='cuda', requires_grad=True)
    rm, rv = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
    y = ops.batch_norm_act(x, gamma, beta, rm, rv, None, False, True, 0.1, 1e-5)
    yd = y.detach().double()
    assert yd.mean(dim=(0, 2, 3)).abs().max() < 1e-5
    assert (yd.var(dim=(0, 2, 3), unbiased=False) - 1).abs().max() < 1e-4
    dy = torch.randn(shape, device='cuda', generator=gen)
    y.backward(dy)
    dx = x.grad.double()
    scale = dy.double().abs().sum(dim=(0, 2, 3))
    assert (dx.sum(dim=(0, 2, 3)).abs() / scale).max() < 1e-6
    assert ((dx * yd).sum(dim=(0, 2, 3)).abs() / scale).max() < 1e-5
    assert (gamma.grad.double() - (dy.double() * yd).sum(dim=(0, 2, 3))).abs().max() < 1e-3 * (dy.double() * yd).sum(dim=(0, 2, 3)).abs().max()
    assert (beta.grad.double() - dy.double().sum(dim=(0, 2, 3))).abs().max() < 1e-3 * dy.double().sum(dim=(0, 2, 3)).abs().max()


X3_CASES = [
    # n, c, k, h, r, stride, dil, bias     (pad = dil * (r - 1) // 2)
    (8, 256, 128, 16, 1, 1, 1, False), (5, 192, 320, 8, 1, 1, 1, False), (64, 512, 128, 32, 1, 1, 1, False), (3, 1024, 256, 16, 1, 1, 1, False),
    (2, 64, 256, 64, 1, 1, 1, False), (2, 256, 64, 16, 1, 1, 1, False), (64, 2048, 512, 16, 1, 1, 1, False),
    (64, 128, 128, 32, 3, 1, 1, False), (64, 256, 272 + 64, 16, 3, 1, 1, True), (64, 128, 512, 16, 3, 1, 2, False), (2, 64, 64, 64, 3, 1, 1, False),
    (4, 512, 512, 16, 3, 1, 2, False), (3, 2048, 272, 16, 3, 1, 1, True), (64, 128, 128, 32, 5, 1, 1, True),
    (8, 128, 128, 64, 3, 2, 1, False), (4, 256, 512, 64, 1, 2, 1, False), (6, 256, 256, 32, 3, 2, 1, False), (2, 128, 256, 32, 3, 1, 4, False),
    (3, 256, 128, 8, 3, 1, 1, False), (64, 256, 256, 16, 3, 1, 1, False),
    # 64 channels on one side: forward / dgrad in half-dead 128-row tiles, the weight gradient on the fp32-MFMA kernel (a 64 x 64-tile x3 kernel measured slower: 60 vs 80 TF)
    (4, 64, 128, 32, 3, 2, 1, False), (3, 64, 64, 8, 3, 1, 1, False), (5, 192, 64, 8, 1, 1, 1, False), (2, 64, 64, 32, 3, 1, 2, False), (64, 64, 64, 64, 3, 1, 1, False),
]


def _x3_covers(n, c, k, h, r, stride, dil):
    """Mirror of fx_fwd_applies / fx_dgrad_applies / fx_wgrad_applies (csrc/p3d_fx.hip) for the square shapes above."""
    ho = (h - 1) // stride + 1
    fwd = c % 16 == 0 and c >= 32 and ho % 4 == 0 and h % 4 == 0 and k >= 96
    dgrad = k % 16 == 0 and k >= 32 and c % 4 == 0 and c >= 96 and ho % 4 == 0 and (h % 4 == 0 if stride == 1 else (h % 8 == 0 and (r == 1 or dil == 1)))
    wgrad = k >= 96 and c >= 96 and (ho * ho) % 16 == 0 and ho % 4 == 0 and h % 4 == 0 and (r == 1 or c % 64 == 0)
    return fwd, dgrad, wgrad


@pytest.mark.parametrize('case', X3_CASES, ids=['n%d_c%d_k%d_h%d_%dx%d_s%d_d%d%s' % (c[0], c[1], c[2], c[3], c[4], c[4], c[5], c[6], '_bias' if c[7] else '') for c in X3_CASES])
def test_x3_kernels_match_fp32_kernels(case, pkg):
    """The default conv path (exact fp32 through the bf16 matrix pipe, csrc/p3d_fx.hip) against float64, next to the fp32-MFMA kernels on the same
    data: forward, data gradient and weight gradient, for 1x1 / 3x3 / 5x5, stride 1 and 2, dilation 1 / 2 / 4, bias, output channel counts that are
    not multiples of the 128-row tile (272, 320, 336) and split-K grids.  Its error is bounded by the fp32 kernel's; where the shape is covered the
    other kernel really ran, where it is not both settings give the same bits."""
    ops = pkg.ops
    n, c, k, h, r, stride, dil, with_bias = case
    pad = dil * (r - 1) // 2
    gen = torch.Generator(device='cuda').manual_seed(11 + c + k)
    F = torch.nn.functional
    x = torch.randn(n, c, h, h, device='cuda', generator=gen) * (torch.rand(n, c, h, h, device='cuda', generator=gen) * 4 - 2).exp2()
    w0 = torch.randn(k, c, r, r, device='cuda', generator=gen) / (c * r * r) ** 0.5
    b0 = torch.randn(k, device='cuda', generator=gen) if with_bias else None
    xd = x.double().requires_grad_(True)
    y_ref = F.conv2d(xd, w0.double(), None if b0 is None else b0.double(), stride, pad, dil)
    dy = torch.randn(y_ref.shape, device='cuda', generator=gen)
    y_ref.backward(dy.double())
    dw_ref = torch.nn.grad.conv2d_weight(x.double(), w0.shape, dy.double(), stride, pad, dil)
    res = {}
    before = ops.set_x3(True)
    try:
        for on in (False, True):
            ops.set_x3(on)
            xr = x.clone().requires_grad_(True)
            w = w0.clone().requires_grad_(True)
            b = None if b0 is None else b0.clone().requires_grad_(True)
            y = ops.conv2d(xr, w, b, stride, pad, dil)
            y.backward(dy)
            res[on] = (y.detach().double(), xr.grad.double(), w.grad.double())
    finally:
        ops.set_x3(before)
    covered = _x3_covers(n, c, k, h, r, stride, dil)
    for i, ref, name in ((0, y_ref.detach(), 'fwd'), (1, xd.grad, 'dgrad'), (2, dw_ref, 'wgrad')):
        scale = ref.abs().max()
        e32, e3 = ((res[False][i] - ref).abs().max() / scale).item(), ((res[True][i] - ref).abs().max() / scale).item()
        # 4e-6 of the largest value: fp32 accumulation over the longest reduction here (128 x 5 x 5 = 3200 terms, six products each) stays below it
        assert e3 < 4e-6 and e3 < 4 * e32 + 2e-7, (name, e32, e3)
        assert torch.equal(res[False][i], res[True][i]) != covered[i], name


def test_x3_accumulates_into_existing_gradients(pkg):
    """accumulate = 1 on the default path: a weight gradient added onto an existing .grad, and a data gradient joined onto the shortcut's (GradJoin)."""
    ops = pkg.ops
    gen = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(4, 256, 16, 16, device='cuda', generator=gen)
    w = (torch.randn(128, 256, 3, 3, device='cuda', generator=gen) / 48).requires_grad_(True)
    dy = torch.randn(4, 128, 16, 16, device='cuda', generator=gen)
    d = ops._desc(x.shape, w.shape, 1, 1, 1, accumulate=1)
    import ctypes
    L = pkg._lib.lib()
    dx = torch.randn(x.shape, device='cuda', generator=gen)
    dx0 = dx.clone()
    ws = ops.workspace(x.device, L.p3d_conv2d_dgrad_workspace_bytes(ctypes.byref(d)))
    pkg._lib.check(L.p3d_conv2d_dgrad(ctypes.byref(d), ops._p(dy), ops._p(w.detach()), None, None, ops._p(dx), ops._p(ws), ws.numel(), ops._stream()), 'dgrad')
    want = torch.nn.grad.conv2d_input(x.shape, w.detach().double(), dy.double(), 1, 1, 1) + dx0.double()
    assert ((dx.double() - want).abs().max() / want.abs().max()).item() < 3e-6
    dw = torch.randn(w.shape, device='cuda', generator=gen)
    dw0 = dw.clone()
    ws = ops.workspace(x.device, L.p3d_conv2d_wgrad_workspace_bytes(ctypes.byref(d)))
    pkg._lib.check(L.p3d_conv2d_wgrad(ctypes.byref(d), ops._p(dy), ops._p(x), None, None, ops._p(dw), ops._p(ws), ws.numel(), ops._stream()), 'wgrad')
    want = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), 1, 1, 1) + dw0.double()
    assert ((dw.double() - want).abs().max() / want.abs().max()).item() < 3e-6


def test_paste_over_and_brightness_contrast_match_reference_golden(pkg):
    """p3d_augment_occlude against the reference's own augment_occluder.paste_over outputs, and the brightness / contrast leg of
    p3d_augment_colour against augment_colour.random_color (no hue / saturation jitter drawn): bit-exact on 0..255 values."""
    g = np.load(golden_path('augment.npz'))
    for m in json.loads(str(g['meta'])):
        n = m['name']
        img = dev(g[n + '.image'].astype(np.float32).transpose(2, 0, 1)[None].copy())
        alpha = g[n + '.alpha'] if m['alpha'] else None
        pkg.augment.paste_over_(img, [g[n + '.occ']], [alpha], g[n + '.center'][None])
        assert np.array_equal(host(img)[0].transpose(1, 2, 0), g[n + '.out'].astype(np.float32)), n
    # a batch with a different occluder per image (and one image left alone) in one launch
    names = ['inside', 'topleft', 'botright']
    batch = dev(np.stack([g[n + '.image'].astype(np.float32).transpose(2, 0, 1) for n in names] + [g['inside.image'].astype(np.float32).transpose(2, 0, 1)]))
    pkg.augment.paste_over_(batch, [g[n + '.occ'] for n in names] + [None], [g[n + '.alpha'] for n in names] + [None],
                            np.stack([g[n + '.center'] for n in names] + [np.zeros(2)]))
    for i, n in enumerate(names):
        assert np.array_equal(host(batch)[i].transpose(1, 2, 0), g[n + '.out'].astype(np.float32)), n
    assert np.array_equal(host(batch)[3].transpose(1, 2, 0), g['inside.image'].astype(np.float32))
    for i, (b, c) in enumerate(g['bc_draws']):
        img = dev(g['bc%d.image' % i].astype(np.float32).transpose(2, 0, 1)[None].copy())
        pkg.ops.augment_colour_(img, dev(np.array([[b, c, 0.0, 1.0]], dtype=np.float32)))
        assert np.array_equal(host(img)[0].transpose(1, 2, 0), g['bc%d.out' % i].astype(np.float32)), i


# ---- pre-split activation images and the image-fed x3 kernels (what p3d_block_* launches inside a residual block) ------------------------------------
def _image_planes(img, n, c, hw):
    """The three bf16 planes of an activation image as float32 [3][N][C][HW] (layout [plane][n][c / 16][hw][16])."""
    raw = img.view(torch.bfloat16).reshape(3, n, c // 16, hw, 16).float()
    return raw.permute(0, 1, 2, 4, 3).reshape(3, n, c, hw)


def test_activation_image_is_an_exact_split(pkg):
    """hi + mid + lo == x bit for bit, each piece a bf16, laid out [n][c/16][pixel][16]; modes 1 / 2 apply relu(bn) / the BatchNorm-backward map first."""
    ops = pkg.ops
    gen = torch.Generator(device='cuda').manual_seed(3)
    n, c, h = 3, 48, 12
    x = torch.randn(n, c, h, h, device='cuda', generator=gen) * torch.logspace(-6, 6, c, device='cuda').view(1, c, 1, 1)
    planes = _image_planes(ops.act_image(x), n, c, h * h)
    back = (planes[0].double() + planes[1].double() + planes[2].double()).float().reshape(x.shape)
    assert torch.equal(back, x)
    assert (planes[1].abs() <= planes[0].abs() * 2.0 ** -7 + 1e-45).all() and (planes[2].abs() <= planes[0].abs() * 2.0 ** -15 + 1e-45).all()
    tab = torch.randn(c, 8, device='cuda', generator=gen)
    x = torch.randn(n, c, h, h, device='cuda', generator=gen)
    c2 = torch.randn(n, c, h, h, device='cuda', generator=gen)
    col = lambda j: tab[:, j].view(1, c, 1, 1)
    want1 = torch.relu(torch.addcmul(col(1), x, col(0)))
    got1 = _image_planes(ops.act_image(x, 1, table=tab), n, c, h * h).double().sum(0).float().reshape(x.shape)
    assert (got1 - want1).abs().max() <= 1e-6 * want1.abs().max()
    for masked in (False, True):
        g = torch.where(torch.addcmul(col(1), c2, col(0)) > 0, x, torch.zeros_like(x)) if masked else x
        want2 = col(4) * g + (col(5) * c2 + col(6))
        got2 = _image_planes(ops.act_image(x, 2, x2=c2, table=tab, masked=masked), n, c, h * h).double().sum(0).float().reshape(x.shape)
        assert (got2 - want2).abs().max() <= 2e-6 * want2.abs().max()


#             N   C    H   K  ks st dil
IMG_CASES = [(2, 64, 16, 64, 3, 1, 1), (2, 128, 16, 272, 3, 1, 1), (3, 256, 16, 128, 1, 1, 1), (2, 128, 32, 128, 3, 2, 1), (2, 128, 16, 128, 3, 1, 2),
             (2, 64, 32, 256, 1, 1, 1), (2, 256, 32, 64, 1, 1, 1), (2, 256, 32, 512, 1, 2, 1), (5, 128, 8, 160, 3, 1, 1), (16, 512, 16, 512, 3, 1, 1),
             # 64 input channels, a multi-tap filter: the two- / three-tap column tiles of the weight gradient (fx_wgrad_kernel<.., TAPS 2 / 3>)
             (2, 64, 16, 128, 3, 1, 1), (3, 64, 32, 48, 3, 1, 1), (2, 64, 32, 64, 3, 2, 1), (2, 64, 16, 64, 5, 1, 1), (2, 64, 16, 32, 3, 1, 2)]


@pytest.mark.parametrize('case', IMG_CASES, ids=['n%d_c%d_h%d_k%d_%dx%d_s%d_d%d' % (c[0], c[1], c[2], c[3], c[4], c[4], c[5], c[6]) for c in IMG_CASES])
def test_image_fed_kernels_match_the_oracle(case, pkg):
    """Forward, data gradient and weight gradient fed by pre-split activation images (AMODE 1 / AIMG / BIMG instances of csrc/p3d_fx.hip) against the float64
    oracle, and against the same kernels fed fp32 tensors (the in-kernel split): the two must agree to rounding of the accumulation order."""
    ops = pkg.ops
    n, c, h, k, ks, st, dil = case
    pad = dil * (ks - 1) // 2
    gen = torch.Generator(device='cuda').manual_seed(c + k + ks + st)
    x = torch.randn(n, c, h, h, device='cuda', generator=gen).requires_grad_(True)
    w = (torch.randn(k, c, ks, ks, device='cuda', generator=gen) / (c * ks * ks) ** 0.5).requires_grad_(True)
    y0 = ops.conv2d(x, w, None, st, pad, dil)
    dy = torch.randn(y0.shape, device='cuda', generator=gen)
    y0.backward(dy)
    x_img, dy_img = ops.act_image(x.detach()), ops.act_image(dy)
    y = ops.conv2d_img('fwd', x.shape, w.detach(), st, pad, dil, x_img=x_img)
    dx = ops.conv2d_img('dgrad', x.shape, w.detach(), st, pad, dil, dy_img=dy_img)
    dw = ops.conv2d_img('wgrad', x.shape, w.detach(), st, pad, dil, dy_img=dy_img, x_img=x_img)
    dw2 = ops.conv2d_img('wgrad', x.shape, w.detach(), st, pad, dil, dy_img=dy_img, x=x.detach())
    torch.cuda.synchronize()
    xh, wh, dyh = host(x), host(w), host(dy)
    want_y = ref.conv2d_fwd(xh, wh, None, st, pad, dil)
    want_dx = ref.conv2d_dgrad(dyh, wh, x.shape, st, pad, dil)
    want_dw = ref.conv2d_wgrad(dyh, xh, w.shape, st, pad, dil)
    for name, got, want, tol in (('fwd', y, want_y, 2e-5), ('dgrad', dx, want_dx, 2e-5), ('wgrad', dw, want_dw, 5e-5), ('wgrad fp32 x', dw2, want_dw, 5e-5)):
        err = np.abs(host(got) - want).max() / np.abs(want).max()
        assert err < tol, (name, err)
    for name, got, other in (('fwd', y, y0.detach()), ('dgrad', dx, x.grad), ('wgrad', dw, w.grad)):
        assert (got - other).abs().max() <= 4e-6 * other.abs().max(), name
    acc = torch.ones_like(x.detach())
    ops.conv2d_img('dgrad', x.shape, w.detach(), st, pad, dil, dy_img=dy_img, accumulate_into=acc)
    assert (acc - 1 - dx).abs().max() <= 4e-6 * dx.abs().max()


@pytest.mark.parametrize('case', [(2, 3, 64, 64, 64), (3, 1, 48, 64, 64), (2, 3, 32, 32, 96), (5, 3, 128, 128, 64), (2, 4, 64, 64, 64)], ids=lambda c: 'n%d_c%d_%dx%d_k%d' % c)
def test_stem_on_the_x3_kernels(case, pkg):
    """conv1 = Conv2d(Cin, K, 7, stride 2, padding 3) (depthnet.py:138) restated as a 4x4 stride-1 convolution over a space-to-depth image of the input
    (p3d_stem_*): forward and weight gradient against the float64 oracle's plain 7x7 stride-2 convolution."""
    n, cin, h, w, k = case
    conv = pkg.nn.Conv2d(cin, k, kernel_size=7, stride=2, padding=3, bias=False).cuda()
    gen = torch.Generator(device='cuda').manual_seed(n + cin + h)
    x = torch.randn(n, cin, h, w, device='cuda', generator=gen)
    assert pkg.ops_block.stem_takes_x3(conv, x)
    y = conv(x)
    assert y.grad_fn is not None and type(y.grad_fn).__name__.startswith('StemConvFn')
    dy = torch.randn(y.shape, device='cuda', generator=gen)
    y.backward(dy)
    pkg.ops.join_side_stream()
    torch.cuda.synchronize()
    xh, wh, dyh = host(x), host(conv.weight), host(dy)
    want_y = ref.conv2d_fwd(xh, wh, None, 2, 3, 1)
    want_dw = ref.conv2d_wgrad(dyh, xh, conv.weight.shape, 2, 3, 1)
    assert np.abs(host(y) - want_y).max() < 2e-5 * np.abs(want_y).max()
    assert np.abs(host(conv.weight.grad) - want_dw).max() < 5e-5 * np.abs(want_dw).max()
    # a second step after a weight update: the cached weight image follows ops.weights_changed()
    with torch.no_grad():
        conv.weight.mul_(0.5)
    y2 = conv(x)
    assert np.abs(host(y2) - 0.5 * want_y).max() < 2e-5 * np.abs(want_y).max()


def test_multi_tap_conv_outside_a_block_takes_image_operands(pkg):
    """nn.Conv2d routes a 3x3 convolution that is not part of a residual block (the `regressor`, depthnet.py:156) through image operands: bias, data gradient,
    weight gradient and bias gradient against the float64 oracle."""
    gen = torch.Generator(device='cuda').manual_seed(9)
    conv = pkg.nn.Conv2d(128, 272, 3, padding=1).cuda()
    x = torch.randn(3, 128, 16, 16, device='cuda', generator=gen).requires_grad_(True)
    assert pkg.ops_block.conv_takes_images(conv, x)
    y = conv(x)
    assert type(y.grad_fn).__name__.startswith('ConvImagesFn')
    dy = torch.randn(y.shape, device='cuda', generator=gen)
    y.backward(dy)
    pkg.ops.join_side_stream()
    torch.cuda.synchronize()
    xh, wh, bh, dyh = host(x), host(conv.weight), host(conv.bias), host(dy)
    want = ref.conv2d_fwd(xh, wh, bh, 1, 1, 1)
    assert np.abs(host(y) - want).max() < 2e-5 * np.abs(want).max()
    want = ref.conv2d_dgrad(dyh, wh, x.shape, 1, 1, 1)
    assert np.abs(host(x.grad) - want).max() < 2e-5 * np.abs(want).max()
    want = ref.conv2d_wgrad(dyh, xh, conv.weight.shape, 1, 1, 1)
    assert np.abs(host(conv.weight.grad) - want).max() < 5e-5 * np.abs(want).max()
    want = dyh.astype(np.float64).sum((0, 2, 3))
    assert np.abs(host(conv.bias.grad) - want).max() < 2e-5 * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(4, 64, 64, 64), (3, 16, 32, 20), (2, 8, 6, 4)])
def test_stem_tail_is_bit_identical_to_batchnorm_relu_maxpool(pkg, shape):
    """maxpool(relu(bn1(x))) as one node (p3d_stem_tail_fwd / bwd, depthnet.py:139-140) computes every value with the expressions and in the order of the
    three-node path: output, input gradient, parameter gradients and running statistics are compared bit for bit (odd window rows / columns at the borders,
    negative gamma, ties between equal maxima included)."""
    ops = pkg.ops
    n, c, h, w = shape
    gen = torch.Generator(device='cuda').manual_seed(h * 7 + w)
    x0 = torch.randn(n, c, h, w, device='cuda', generator=gen)
    x0 = (x0 * 2).round() / 2                                 # values on a coarse grid: plenty of exact ties inside the pooling windows
    dy = torch.randn(n, c, h // 2, w // 2, device='cuda', generator=gen)
    res = {}
    for fused in (True, False):
        torch.manual_seed(1)
        bn = pkg.nn.BatchNorm2d(c).cuda().train()
        pool = pkg.nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        with torch.no_grad():
            bn.weight.uniform_(-1.0, 1.5); bn.bias.normal_(0, 0.3); bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5)
        x = x0.clone().requires_grad_(True)
        assert ops.stem_tail_usable(x, bn, pool)
        y = ops.stem_tail(x, bn) if fused else pool(bn(x, relu=True))
        y.backward(dy)
        torch.cuda.synchronize()
        res[fused] = (y.detach().clone(), x.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(), bn.running_mean.clone(), bn.running_var.clone(),
                      bn.num_batches_tracked.clone())
    for a, b, name in zip(res[True], res[False], ('y', 'dx', 'dgamma', 'dbeta', 'running_mean', 'running_var', 'num_batches_tracked')):
        assert torch.equal(a, b), name


def test_extra_channel_stem_forward_backward(pkg):
    """`-extra_channel` (resnet.py:142): the legacy network's stem takes 4 input channels.  conv1 -> bn1 -> ReLU -> max pool of that network, forward and backward
    (weight gradient of the 4-channel conv1, d gamma / d beta of bn1), against the float64 oracle's chain of the same four operators."""
    args = pkg.opts.parse(['-model', 'resnet18', '-suffix', 't', '-data_name', 'h36m', '-save_path', '/tmp/p3d', '-criterion', 'SmoothL1', '-num_joints', '17',
                           '-side_in', '64', '-extra_channel'])
    torch.manual_seed(3)
    model = pkg.resnet.resnet18(args).cuda().train()
    assert tuple(model.conv1.weight.shape) == (64, 4, 7, 7)
    with torch.no_grad():
        model.bn1.weight.uniform_(0.5, 1.5)
        model.bn1.bias.uniform_(-0.3, 0.3)
    gen = torch.Generator(device='cuda').manual_seed(11)
    x = torch.randn(3, 4, 64, 64, device='cuda', generator=gen)
    stem = __import__('importlib').import_module(pkg.__name__ + '._trunk').stem
    out = stem(model.conv1, model.bn1, model.maxpool, x)
    dout = torch.randn(out.shape, device='cuda', generator=gen)
    out.backward(dout)
    pkg.ops.join_side_stream()
    torch.cuda.synchronize()
    xh, wh, gh, bh, dh = host(x), host(model.conv1.weight), host(model.bn1.weight), host(model.bn1.bias), host(dout)
    c = ref.conv2d_fwd(xh, wh, None, 2, 3, 1)
    y, mean, invstd, _, _ = ref.bn_train_fwd(c, gh, bh)
    a = ref.relu_fwd(y)
    want, idx = ref.maxpool3x3s2_fwd(a)
    assert np.abs(host(out) - want).max() < 2e-5 * np.abs(want).max()
    da = ref.maxpool3x3s2_bwd(dh, idx, a.shape)
    dy = ref.relu_bwd(da, a)
    dc, dgamma, dbeta = ref.bn_train_bwd(dy, c, mean, invstd, gh)
    dw = ref.conv2d_wgrad(dc, xh, wh.shape, 2, 3, 1)
    assert np.abs(host(model.bn1.weight.grad) - dgamma).max() < 5e-5 * np.abs(dgamma).max()
    assert np.abs(host(model.bn1.bias.grad) - dbeta).max() < 5e-5 * np.abs(dbeta).max()
    assert np.abs(host(model.conv1.weight.grad) - dw).max() < 1e-4 * np.abs(dw).max()


def test_large_filter_weight_gradient_with_deep_split(pkg):
    """An 11x11 filter whose weight gradient is cut into more than 16 slabs (forced): the one-launch slab sum needs 17 * 121 * 16 floats of LDS (> 64 KB), so the call
    must take the fold + reduce pair instead of a launch that cannot start; against the float64 oracle."""
    ops = pkg.ops
    L = pkg._lib.lib()
    rng = np.random.default_rng(5)
    x = rng.standard_normal((10, 64, 16, 16)).astype(np.float32)
    wt = (rng.standard_normal((96, 64, 11, 11)) / np.sqrt(64 * 121)).astype(np.float32)
    dy = rng.standard_normal((10, 96, 16, 16)).astype(np.float32)
    xt, wtt = dev(x).requires_grad_(True), dev(wt).requires_grad_(True)
    L.p3d_fx_tune(0, 20)
    try:
        y = ops.conv2d(xt, wtt, None, 1, 5, 1)
        y.backward(dev(dy))
        ops.join_side_stream()
        torch.cuda.synchronize()
    finally:
        L.p3d_fx_tune(0, 0)
    assert relerr(host(wtt.grad), ref.conv2d_wgrad(dy, x, wt.shape, 1, 5, 1)) < 2e-5


@pytest.mark.parametrize('case', [(3, 1, 64, 64), (2, 1, 128, 96)], ids=lambda c: 'n%d_c%d_%dx%d' % c)
def test_partial_conv_stem_on_the_restated_kernels(case, pkg):
    """The 1-channel PartialConv stem of the partial families (partial_depthnet.py:177: PartialConv(1, 64, 7, stride 2, padding 3); partial_conv.py:32-57) on the
    restated stem kernels: mask_in is multiplied into the space-to-depth image, mult scales the result in the forward epilogue and dy as the weight gradient fetches
    it.  Forward, mask_out and the weight gradient against the float64 oracle; a general 0/1 mask (not only x != 0) and pixels whose whole window is masked."""
    n, cin, h, w = case
    conv = pkg.partial_conv.PartialConv(cin, 64, kernel_size=7, stride=2, padding=3, bias=False).cuda()
    gen = torch.Generator(device='cuda').manual_seed(n + h)
    x = torch.randn(n, cin, h, w, device='cuda', generator=gen)
    mask = (torch.rand(n, 1, h, w, device='cuda', generator=gen) > 0.35).float()
    mask[0, 0, :24, :24] = 0.0                                                    # windows without a single valid pixel: mult = 0 there
    assert pkg.ops_block.stem_takes_x3(conv, x, masked=True)
    y, mask_out = conv(x, mask)
    assert type(y.grad_fn).__name__.startswith('StemConvFn')
    dy = torch.randn(y.shape, device='cuda', generator=gen)
    y.backward(dy)
    pkg.ops.join_side_stream()
    torch.cuda.synchronize()
    xh, mh, wh, dyh = host(x), host(mask), host(conv.weight), host(dy)
    want_y, want_mo, mult = ref.partial_conv_fwd(xh, mh, wh, None, 2, 3, 1)
    _, want_dw = ref.partial_conv_bwd(dyh, xh, mh, wh, mult, 2, 3, 1)
    assert np.array_equal(host(mask_out), want_mo)
    assert np.abs(host(y) - want_y).max() < 2e-5 * np.abs(want_y).max()
    assert np.all(host(y)[0, :, :9, :9] == 0.0)                                   # fully masked windows give exact zeros
    assert np.abs(host(conv.weight.grad) - want_dw).max() < 5e-5 * np.abs(want_dw).max()

"""Pin the oracle (oracle/np_ops.py, oracle/np_net.py) against vectors produced by the real
reference (tests/golden/make_golden.py).  CPU only."""
import json

import numpy as np
import pytest

from conftest import golden_path
from oracle import np_ops as ops
from oracle import np_net


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_partial_conv_matches_reference():
    g = np.load(golden_path('partial_conv.npz'))
    for m in json.loads(str(g['meta'])):
        n = m['name']
        b = g[n + '.b'] if m['bias'] else None
        y, mo, mult = ops.partial_conv_fwd(g[n + '.x'], g[n + '.mask'], g[n + '.w'], b, m['stride'], m['pad'], m['dil'])
        assert np.array_equal(mo, g[n + '.mask_out']), n
        assert rel(y, g[n + '.y']) < 2e-6, n
        if not m['bias']:
            dx, dw = ops.partial_conv_bwd(g[n + '.dy'], g[n + '.x'], g[n + '.mask'], g[n + '.w'], mult,
                                          m['stride'], m['pad'], m['dil'])
            assert np.abs(dx - g[n + '.dx']).max() < 2e-6 * max(np.abs(g[n + '.dx']).max(), 1.0), n
            assert np.abs(dw - g[n + '.dw']).max() < 2e-6 * max(np.abs(g[n + '.dw']).max(), 1.0), n


def test_head_matches_reference():
    g = np.load(golden_path('head.npz'))
    for m in json.loads(str(g['meta'])):
        n = m['name']
        a = (m['depth'], m['num_joints'], m['height'], m['width'])
        coords = ops.softargmax3d_fwd(g[n + '.z'], *a, m['depth_range'])
        assert rel(coords, g[n + '.coords']) < 2e-6, n
        dz = ops.softargmax3d_bwd(g[n + '.dc'], g[n + '.z'], *a, m['depth_range'])
        assert rel(dz, g[n + '.dz']) < 5e-5, n
        if n == 'tiny':
            assert rel(ops.to_heatmap(g[n + '.z'], *a), g[n + '.heat']) < 2e-6


STEP_CASES = ['depth_r18_b2', 'depth_r18_odd_b1', 'depthonly_r18_b2', 'fusion_r18_b2', 'partial_r18_b2',
              'depth_r50_b2', 'fusion_r50_b1', 'partial_r50_b1', 'pfusion_r18_b2', 'pfusion_r50_b1',
              'depth_r18_s8_b2', 'depth_r18_s32_b2', 'depth_r18_s4_b1']      # the other stage geometries of depthnet.py:130-136


def net_stride(meta):
    return int(meta['extra'][meta['extra'].index('-stride') + 1]) if '-stride' in meta['extra'] else 16


@pytest.mark.parametrize('case', STEP_CASES)
def test_train_step_matches_reference(case, synth):
    g = np.load(golden_path('step_%s.npz' % case))
    meta = json.loads(str(g['meta']))
    with open(golden_path('state_keys.json')) as f:
        inv = json.load(f)
    tag = meta['family'] + ('_depth_only' if (meta['family'] == 'depthnet' and '-depth_only' in meta['extra']) else '')
    shapes = inv[tag + '.' + meta['model']]['state']
    sd = synth.det_state_dict(shapes, 0)
    names = meta['names']
    state = None
    acc = np.float64       # the golden side is the reference's own fp32; R50 with B<=2 carries ~1e-4 of fp32 noise in z
    for it in range(meta['iters']):
        c, d, tc, tv = synth.make_batch(meta['batch'], side=meta['side'], rank=0, step=it, invalid_frac=meta['invalid_frac'])
        out = np_net.train_step(sd, c, d, tc, tv, family=meta['family'], model=meta['model'],
                                depth_only='-depth_only' in meta['extra'], lr=meta['lr'], adam_state=state,
                                step=it + 1, acc=acc, stride=net_stride(meta))
        assert abs(out['loss'] - g['losses'][it]) < 2e-4 * abs(g['losses'][it]), (it, out['loss'], g['losses'][it])
        assert abs(out['clip_total'] - g['clip_total'][it]) < 2e-3 * g['clip_total'][it]
        spec_sel = out['spec_cam'].reshape(-1, 3)[tv.reshape(-1)]
        assert rel(spec_sel, g['spec_sel_%d' % it]) < 1e-4
        if it == 0:
            assert rel(out['z'][0, :, 3, 5], g['z_first_slice']) < (2e-4 if meta['model'] == 'resnet18' else 1e-3)
        sd, state = out['new_sd'], out['adam_state']
    assert rel(out['z'][0, :, 3, 5], g['z_last_slice']) < 1e-3
    gn = np.array([np.linalg.norm(out['grads'][n].astype(np.float64)) for n in names])
    assert np.abs(gn - g['grad_norms']).max() < 2e-3 * g['grad_norms'].max()
    assert np.all(np.abs(gn - g['grad_norms']) < 2e-2 * g['grad_norms'] + 1e-4 * g['grad_norms'].max())
    pn = np.array([np.linalg.norm(sd[n].astype(np.float64)) for n in names])
    assert np.abs(pn - g['param_norms']).max() < 1e-5 * g['param_norms'].max()
    ps = np.array([sd[n].reshape(-1)[g['sample_idx'][i]] for i, n in enumerate(names)])
    assert np.abs(ps - g['param_samples']).max() < 3e-5     # Adam moves each weight by <= lr = 1e-5 per step
    bn = np.array([np.linalg.norm(sd[k].astype(np.float64)) for k in meta['buffer_names']])
    assert np.abs(bn - g['buffer_norms']).max() < 1e-4 * max(g['buffer_norms'].max(), 1.0)


def test_learn_rate_schedule():
    # depth_train.py:621-638
    assert ops.adapt_learn_rate(1, 5e-5) == pytest.approx(1e-5)
    assert ops.adapt_learn_rate(2, 5e-5) == pytest.approx(5e-5)
    assert ops.adapt_learn_rate(16, 5e-5) == pytest.approx(1e-5)
    assert ops.adapt_learn_rate(21, 5e-5) == pytest.approx(2e-6)
    assert ops.adapt_learn_rate(26, 5e-5) == pytest.approx(4e-7)


@pytest.mark.parametrize('case', ['depth_r18_b2', 'fusion_r18_b2', 'partial_r18_b2', 'depth_r50_b2'])
def test_torch_port_matches_reference(case, synth):
    """oracle/torch_port.py (the cpu_baseline of bench.py) against the same reference vectors."""
    from oracle.torch_port import TorchPort
    g = np.load(golden_path('step_%s.npz' % case))
    meta = json.loads(str(g['meta']))
    with open(golden_path('state_keys.json')) as f:
        inv = json.load(f)
    tag = meta['family'] + ('_depth_only' if (meta['family'] == 'depthnet' and '-depth_only' in meta['extra']) else '')
    sd = synth.det_state_dict(inv[tag + '.' + meta['model']]['state'], 0)
    port = TorchPort(sd, family=meta['family'], model=meta['model'])
    for it in range(meta['iters']):
        c, d, tc, tv = synth.make_batch(meta['batch'], side=meta['side'], rank=0, step=it, invalid_frac=meta['invalid_frac'])
        out = port.train_step(c, d, tc, tv, depth_only='-depth_only' in meta['extra'], lr=meta['lr'])
        assert abs(out['loss'] - g['losses'][it]) < 1e-4 * abs(g['losses'][it])
        assert abs(out['clip_total'] - g['clip_total'][it]) < 1e-3 * g['clip_total'][it]
        assert rel(out['spec_cam'].reshape(-1, 3)[tv.reshape(-1)], g['spec_sel_%d' % it]) < 1e-4
    st = port.state()
    ps = np.array([st[n].reshape(-1)[g['sample_idx'][i]] for i, n in enumerate(meta['names'])])
    assert np.abs(ps - g['param_samples']).max() < 3e-5


def test_augment_restatements_match_reference(pkg):
    """oracle.np_ops.paste_over / brightness_contrast and the host planner augment.plan_paste against the reference's own
    augment_occluder.paste_over and augment_colour.random_color (brightness / contrast leg) outputs: bit-exact (uint8)."""
    g = np.load(golden_path('augment.npz'))
    for m in json.loads(str(g['meta'])):
        n = m['name']
        alpha = g[n + '.alpha'] if m['alpha'] else None
        out = ops.paste_over(g[n + '.occ'], g[n + '.image'].copy(), alpha, g[n + '.center'])
        assert np.array_equal(out, g[n + '.out']), n
        dy, dx, sy, sx, h, w = pkg.augment.plan_paste(g[n + '.occ'].shape, g[n + '.image'].shape, g[n + '.center'])
        changed = np.argwhere((g[n + '.out'] != g[n + '.image']).any(axis=2))
        assert h > 0 and w > 0
        assert changed[:, 0].min() >= dy and changed[:, 0].max() < dy + h and changed[:, 1].min() >= dx and changed[:, 1].max() < dx + w, n
    for i, (b, c) in enumerate(g['bc_draws']):
        assert np.array_equal(ops.brightness_contrast(g['bc%d.image' % i], b, c), g['bc%d.out' % i]), i


def test_sampled_conv_oracle_equals_dense_oracle():
    """conv2d_*_at / conv2d_wgrad_block (the checkers of the full-size GPU parity tests) against the dense oracle pinned above."""
    rng = np.random.default_rng(0)
    for (n, c, h, k, ks, st, pad, dil) in [(3, 5, 11, 7, 3, 2, 1, 1), (2, 4, 9, 6, 3, 1, 2, 2), (2, 6, 8, 5, 1, 2, 0, 1), (2, 3, 12, 4, 7, 2, 3, 1)]:
        x = rng.standard_normal((n, c, h, h)).astype(np.float32)
        w = rng.standard_normal((k, c, ks, ks)).astype(np.float32)
        b = rng.standard_normal(k).astype(np.float32)
        y = ops.conv2d_fwd(x, w, b, st, pad, dil)
        dy = rng.standard_normal(y.shape).astype(np.float32)
        idx = np.stack([rng.integers(0, d, 300) for d in y.shape], 1)
        assert np.allclose(ops.conv2d_fwd_at(x, w, b, st, pad, dil, idx), y[tuple(idx.T)], atol=1e-4)
        idx = np.stack([rng.integers(0, d, 300) for d in x.shape], 1)
        assert np.allclose(ops.conv2d_dgrad_at(dy, w, x.shape, st, pad, dil, idx), ops.conv2d_dgrad(dy, w, x.shape, st, pad, dil)[tuple(idx.T)], atol=1e-4)
        ksel, csel = np.array([0, k - 1, 2]), np.array([c - 1, 0])
        assert np.allclose(ops.conv2d_wgrad_block(dy, x, ksel, csel, ks, ks, st, pad, dil), ops.conv2d_wgrad(dy, x, w.shape, st, pad, dil)[ksel][:, csel], atol=1e-4)

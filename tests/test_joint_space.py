"""Legacy joint-space path (`get_recon_cam` LSQ head; train.py:54-145): oracle and kernels against tests/golden/joint.npz,
which holds the reference's own get_recon_cam / get_deter_cam / mat_utils outputs and one unmodified Trainer.joint_train iteration."""
import json

import numpy as np
import pytest
import torch

from conftest import golden_path
from oracle import np_ops


def rel(a, b):
    return np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30)


def test_oracle_recon_and_mat_head_match_reference(pkg):
    g = np.load(golden_path('joint.npz'))
    recon, cache = np_ops.recon_cam(g['recon.spec_mat'], g['recon.relat'], g['recon.intr'])
    assert rel(recon, g['recon.out']) < 2e-5                                      # the reference solves the 3x3 system in fp32
    assert rel(recon, g['recon.deter']) < 1e-7                                    # its numpy twin is float64
    dspec, drelat = np_ops.recon_cam_bwd(g['recon.drecon'], cache)
    assert rel(dspec, g['recon.dspec_mat']) < 2e-3 and rel(drelat, g['recon.drelat']) < 2e-5
    # host twin of the package
    assert rel(pkg.utils.get_deter_cam(g['recon.spec_mat'], g['recon.relat'], g['recon.intr'], g['recon.valid']), g['recon.deter']) < 1e-7
    with pytest.raises(AssertionError):
        pkg.utils.get_deter_cam(g['recon.spec_mat'], g['recon.relat'], g['recon.intr'], np.zeros_like(g['recon.valid']))
    coords, heat = np_ops.softargmax2d(g['mat.z'], 128)
    assert rel(coords, g['mat.coords']) < 2e-6
    assert rel(np_ops.softargmax2d_bwd(g['mat.dc'], heat, coords, 128), g['mat.dz']) < 2e-5
    stats = json.loads(str(g['mat.stats']))
    mine = pkg.mat_utils.analyze(g['mat.coords'], g['mat.true'], g['recon.valid'][:2], 128)
    assert mine['batch_size'] == stats['batch_size'] and mine['mat_mean'] == pytest.approx(stats['mat_mean'], rel=1e-6)
    assert mine['score_oks'] == pytest.approx(stats['score_oks'], rel=1e-6)
    assert pkg.mat_utils.parse_epoch([mine, mine])['score_oks'] == pytest.approx(stats['score_oks'], rel=1e-6)
    # finite-difference check of the oracle's own backward
    eps = 1e-4
    sm = g['recon.spec_mat'].astype(np.float64)
    for idx in [(0, 3, 1), (2, 9, 0)]:
        hi, lo = sm.copy(), sm.copy()
        hi[idx] += eps
        lo[idx] -= eps
        fd = ((np_ops.recon_cam(hi, g['recon.relat'], g['recon.intr'])[0] - np_ops.recon_cam(lo, g['recon.relat'], g['recon.intr'])[0]) * g['recon.drecon']).sum() / (2 * eps)
        assert fd == pytest.approx(dspec[idx], rel=1e-5)


@pytest.mark.gpu
def test_recon_cam_and_mat_head_kernels(pkg):
    g = np.load(golden_path('joint.npz'))
    dev = 'cuda'
    sm = torch.from_numpy(g['recon.spec_mat']).to(dev).requires_grad_(True)
    rc = torch.from_numpy(g['recon.relat']).to(dev).requires_grad_(True)
    recon = pkg.utils.get_recon_cam(sm, rc, torch.from_numpy(g['recon.intr']).to(dev), torch.from_numpy(g['recon.valid']).to(dev))
    recon.backward(torch.from_numpy(g['recon.drecon']).to(dev))
    want, cache = np_ops.recon_cam(g['recon.spec_mat'], g['recon.relat'], g['recon.intr'])
    dspec, drelat = np_ops.recon_cam_bwd(g['recon.drecon'], cache)
    assert rel(recon.detach().cpu().numpy(), want) < 1e-6 and rel(recon.detach().cpu().numpy(), g['recon.out']) < 2e-5
    assert rel(sm.grad.cpu().numpy(), dspec) < 1e-5 and rel(rc.grad.cpu().numpy(), drelat) < 1e-6
    assert rel(sm.grad.cpu().numpy(), g['recon.dspec_mat']) < 2e-3 and rel(rc.grad.cpu().numpy(), g['recon.drelat']) < 2e-5
    with pytest.raises(AssertionError):
        pkg.utils.get_recon_cam(sm, rc, torch.from_numpy(g['recon.intr']).to(dev), torch.zeros(3, 17, dtype=torch.bool, device=dev))
    z = torch.from_numpy(g['mat.z']).to(dev).requires_grad_(True)
    coords = pkg.mat_utils.decode(pkg.mat_utils.to_heatmap(z, 17, 8, 8), 128)
    coords.backward(torch.from_numpy(g['mat.dc']).to(dev))
    assert coords.shape == (2, 17, 2) and rel(coords.detach().cpu().numpy(), g['mat.coords']) < 2e-6
    assert rel(z.grad.cpu().numpy(), g['mat.dz']) < 2e-5
    # masked criterion against the oracle, all three criteria
    rng = np.random.Generator(np.random.PCG64(2))
    pred, target = rng.standard_normal((4, 17, 2)).astype(np.float32) * 2, rng.standard_normal((4, 17, 2)).astype(np.float32)
    valid = rng.random((4, 17)) > 0.3
    for crit in ('SmoothL1', 'L1', 'MSE'):
        p = torch.from_numpy(pred).to(dev).requires_grad_(True)
        loss = pkg.ops.masked_loss(p, torch.from_numpy(target).to(dev), torch.from_numpy(valid).to(dev), crit)
        (loss * 3.0).backward()
        want_loss, want_grad = np_ops.masked_loss(pred, target, valid, crit)
        assert float(loss.detach()) == pytest.approx(want_loss, rel=1e-6) and rel(p.grad.cpu().numpy(), 3.0 * want_grad) < 1e-6


@pytest.mark.gpu
def test_joint_train_step_matches_reference(pkg, synth):
    """One Trainer.joint_train iteration (-joint_space -do_track, epoch 2: loss = (cam + mat) / 2 + recon) against the reference's."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('make_golden_inputs', os.path.join(os.path.dirname(__file__), 'golden', 'joint_inputs.py'))
    inputs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(inputs)
    g = np.load(golden_path('joint.npz'))
    meta = json.loads(str(g['step.meta']))
    args = pkg.opts.parse(['-model', 'resnet18', '-suffix', 't', '-data_name', 'h36m', '-save_path', '/tmp/p3d', '-criterion', 'SmoothL1', '-num_joints', '17',
                           '-side_in', '128', '-joint_space', '-do_track'])
    args.thresh_solid, args.thresh_close, args.thresh_rough = 40.0, 80.0, 150.0
    model = pkg.resnet.resnet18(args)
    sd = model.state_dict()
    det = synth.det_state_dict({k: tuple(v.shape) for k, v in sd.items()}, 0)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in det.items()})
    model = model.cuda()
    trainer = pkg.train.Trainer(args, model, pkg.utils.get_info())
    trainer.verbose = False
    assert [n for n, _ in model.named_parameters()] == meta['names']
    batch = [torch.from_numpy(a) for a in inputs.joint_batch(synth, 2, 128, 0)]
    record = trainer.train(2, [tuple(batch)])
    assert trainer.optimizer.param_groups[0]['lr'] == pytest.approx(meta['lr'])
    for key in ('cam_train_loss', 'mat_train_loss', 'recon_train_loss'):
        assert record[key] == pytest.approx(meta['record'][key], rel=1e-3), key
    assert trainer.optimizer.total_norm() == pytest.approx(meta['clip_total'], rel=5e-3)
    grads = [p.grad.detach().cpu().numpy() for p in trainer.list_params]
    gn = np.array([np.linalg.norm(x.astype(np.float64)) for x in grads])
    assert np.abs(gn - g['step.grad_norms']).max() < 5e-3 * g['step.grad_norms'].max()
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    pn = np.array([np.linalg.norm(sd[n].astype(np.float64)) for n in meta['names']])
    assert np.abs(pn - g['step.param_norms']).max() < 1e-5 * g['step.param_norms'].max()
    ps = np.array([sd[n].reshape(-1)[g['step.sample_idx'][i]] for i, n in enumerate(meta['names'])])
    assert np.abs(ps - g['step.param_samples']).max() < 2e-4        # one Adam step moves a weight by <= lr = 1e-4 here
    # evaluation pass of the joint-space trainer runs and reports both heads (+ the DETER track)
    c, cam, mat, tv, intr = batch
    test_rec = trainer.test(2, [(c, cam, mat, torch.eye(3).repeat(2, 1, 1), tv, intr)])
    assert {'cam_test_loss', 'mat_test_loss', 'score_oks', 'mat_mean', 'score_pck', 'recon_score_pck'} <= set(test_rec)

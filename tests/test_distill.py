"""Distillation ("privileged information") against vectors from the reference's own Trainer.distill /
distill_train / utils.get_attention."""
import json

import numpy as np
import pytest
import torch

from conftest import golden_path


def test_attention_map_matches_reference(pkg):
    g = np.load(golden_path('distill.npz'))
    att = pkg.utils.get_attention(80, 16, g['coords'], True)
    assert att.shape == (1, 5, 5) and np.abs(att - g['att']).max() < 1e-6
    assert np.array_equal(pkg.utils.get_attention(80, 16, g['coords'], False), np.ones((1, 5, 5)))


def test_alpha_schedule(pkg):
    # depth_train.py:641-647 with the opts defaults alpha_init = alpha_dest = 0.1, alpha_span = 10
    args = pkg.opts.parse(['-model', 'resnet18', '-suffix', 't', '-data_name', 'h36m', '-save_path', '/tmp/p3d', '-criterion', 'SmoothL1',
                           '-num_joints', '17', '-side_in', '128', '-do_teach', '-do_fusion', '-alpha_init', '0.5', '-alpha_dest', '0.1', '-alpha_span', '5'])
    model = pkg.depthnet.resnet18(args, False)
    tr = pkg.depth_train.Trainer(args, model, pkg.utils.get_info())
    want = np.linspace(0.5, 0.1, 5)
    assert [tr.get_dist_weight(e) for e in range(1, 6)] == pytest.approx(list(want))
    assert tr.get_dist_weight(6) == 0.1 and tr.get_dist_weight(30) == 0.1


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['l2', 'sigmoid', 'bce'])
def test_distill_loss_matches_reference(mode, pkg):
    g = np.load(golden_path('distill.npz'))
    s = torch.from_numpy(g['s']).cuda().requires_grad_(True)
    weighted, raw = pkg.ops.distill_loss(torch.from_numpy(g['t']).cuda(), s, torch.from_numpy(g['a']).cuda(), mode, weight=1.0)
    weighted.backward()
    assert float(raw) == pytest.approx(float(g[mode + '.loss']), rel=1e-5)
    ref = g[mode + '.ds']
    assert np.abs(s.grad.cpu().numpy() - ref).max() < 1e-5 * max(np.abs(ref).max(), 1e-12)
    # weight scales the gradient, not the reported loss
    s2 = torch.from_numpy(g['s']).cuda().requires_grad_(True)
    w2, raw2 = pkg.ops.distill_loss(torch.from_numpy(g['t']).cuda(), s2, torch.from_numpy(g['a']).cuda(), mode, weight=0.25, unit_grad=True)
    w2.backward()
    assert float(raw2) == pytest.approx(float(raw)) and float(w2) == pytest.approx(0.25 * float(raw), rel=1e-6)
    assert np.abs(s2.grad.cpu().numpy() - 0.25 * ref).max() < 1e-5 * max(np.abs(ref).max(), 1e-12)


@pytest.mark.gpu
def test_distill_train_step_matches_reference(pkg):
    g = np.load(golden_path('distill.npz'))
    args = pkg.opts.parse(['-model', 'resnet18', '-suffix', 't', '-data_name', 'h36m', '-save_path', '/tmp/p3d', '-criterion', 'SmoothL1',
                           '-num_joints', '17', '-side_in', '128', '-do_teach', '-do_fusion'])
    student = pkg.depthnet.resnet18(args, False)
    teacher = pkg.fusionnet.resnet18(args, False)
    for net, seed in ((student, 0), (teacher, 1)):
        det = pkg.synth.det_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed)
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in det.items()})
    trainer = pkg.depth_train.Trainer(args, student.cuda(), pkg.utils.get_info())
    trainer.set_teacher(teacher.cuda())
    trainer.verbose = False
    c, d, tc, tv = pkg.synth.make_batch(2, side=128, rank=11, step=0)
    batch = tuple(torch.from_numpy(x) for x in (c, d, tc, tv, g['step.att']))
    record = trainer.train(1, [batch])
    want = json.loads(str(g['step.record']))
    assert trainer.get_dist_weight(1) == pytest.approx(float(g['step.alpha']))
    assert record['cam_train_loss'] == pytest.approx(want['cam_train_loss'], rel=1e-3)
    assert record['dist_train_loss'] == pytest.approx(want['dist_train_loss'], rel=1e-3)
    names = json.loads(str(g['step.names']))
    sd = {k: v.detach().cpu().numpy() for k, v in student.state_dict().items()}
    pn = np.array([np.linalg.norm(sd[n].astype(np.float64)) for n in names])
    assert np.abs(pn - g['step.param_norms']).max() < 1e-5 * g['step.param_norms'].max()
    ps = np.array([sd[n].reshape(-1)[g['step.sample_idx'][i]] for i, n in enumerate(names)])
    assert np.abs(ps - g['step.param_samples']).max() < 3e-5


@pytest.mark.gpu
def test_semi_teach_step_matches_reference(pkg):
    """-semi_teach: a batch of unlabelled pairs adds its distillation loss (depth_train.py:132-153,222-230); one iteration against
    the reference's own distill_train with the same labelled and unlabelled batches."""
    g = np.load(golden_path('distill_semi.npz'))
    args = pkg.opts.parse(['-model', 'resnet18', '-suffix', 't', '-data_name', 'h36m', '-save_path', '/tmp/p3d', '-criterion', 'SmoothL1',
                           '-num_joints', '17', '-side_in', '128', '-do_teach', '-do_fusion', '-semi_teach', '-semi_batch', '2',
                           '-synthetic', '1', '-workers', '0'])
    student = pkg.depthnet.resnet18(args, False)
    teacher = pkg.fusionnet.resnet18(args, False)
    for net, seed in ((student, 0), (teacher, 1)):
        det = pkg.synth.det_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed)
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in det.items()})
    trainer = pkg.depth_train.Trainer(args, student.cuda(), pkg.utils.get_info())
    assert trainer.semi_loader is not None and args.data_name == 'h36m' and args.batch_size == 64        # the caller's namespace is left alone
    c, d, tc, tv = pkg.synth.make_batch(2, side=128, rank=12, step=0)
    trainer.semi_loader = [tuple(torch.from_numpy(x) for x in (c, d, tc, tv, g['semi_att']))]            # the golden's unlabelled batch
    trainer.semi_worker = iter(trainer.semi_loader)
    trainer.set_teacher(teacher.cuda())
    trainer.verbose = False
    c, d, tc, tv = pkg.synth.make_batch(2, side=128, rank=11, step=0)
    record = trainer.train(1, [tuple(torch.from_numpy(x) for x in (c, d, tc, tv, g['att']))])
    want = json.loads(str(g['record']))
    assert record['cam_train_loss'] == pytest.approx(want['cam_train_loss'], rel=1e-3)
    assert record['dist_train_loss'] == pytest.approx(want['dist_train_loss'], rel=1e-3)
    names = json.loads(str(g['names']))
    sd = {k: v.detach().cpu().numpy() for k, v in student.state_dict().items()}
    pn = np.array([np.linalg.norm(sd[n].astype(np.float64)) for n in names])
    assert np.abs(pn - g['param_norms']).max() < 1e-5 * g['param_norms'].max()
    ps = np.array([sd[n].reshape(-1)[g['sample_idx'][i]] for i, n in enumerate(names)])
    assert np.abs(ps - g['param_samples']).max() < 3e-5
    bn = np.array([np.linalg.norm(sd[k].astype(np.float64)) for k in sd if k not in names])
    assert np.abs(bn - g['buffer_norms']).max() < 1e-4 * max(g['buffer_norms'].max(), 1.0)       # two student forwards -> two BN updates

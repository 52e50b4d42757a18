"""Evaluation path: host metrics against the reference's utils.analyze / parse_epoch, and Trainer.test
(frozen-BN forward on the HIP kernels) against the record the reference's vanilla_test produced."""
import json

import numpy as np
import pytest
import torch

from conftest import golden_path


def test_metrics_match_reference(pkg):
    g = np.load(golden_path('eval.npz'))
    thresh = json.loads(str(g['thresh']))
    info = pkg.utils.get_info()
    stats = []
    for i in range(3):
        st = pkg.utils.analyze(g['an%d.spec' % i], g['an%d.true' % i], g['an%d.val' % i], info.mirror, thresh)
        want = json.loads(str(g['an%d.stats' % i]))
        assert set(st) == set(want)
        for k, v in want.items():
            assert float(st[k]) == pytest.approx(v, rel=1e-6, abs=1e-9), (i, k)
        stats.append(st)
    epoch = pkg.utils.parse_epoch(stats)
    want = json.loads(str(g['epoch']))
    assert set(epoch) == set(want)
    for k, v in want.items():
        assert epoch[k] == pytest.approx(v, rel=1e-6, abs=1e-9), k


@pytest.mark.gpu
def test_trainer_test_matches_reference(pkg, tmp_path):
    g = np.load(golden_path('eval.npz'))
    meta = tmp_path / 'metadata.json'
    meta.write_text(json.dumps(dict(loader=dict(h36m='depth_datasets'), no_depth=dict(h36m=False),
                                    thresholds=dict(h36m=json.loads(str(g['thresh']))), root=dict(h36m=str(tmp_path)))))
    args = pkg.opts.parse(['-model', 'resnet18', '-suffix', 't', '-data_name', 'h36m', '-save_path', '/tmp/p3d', '-criterion', 'SmoothL1',
                           '-num_joints', '17', '-side_in', '256', '-metadata', str(meta)])
    model, _ = pkg.depth_main.create_model(args)
    det = pkg.synth.det_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 0)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in det.items()})
    trainer = pkg.depth_train.Trainer(args, model.cuda(), pkg.utils.get_info())
    trainer.verbose = False
    batches = []
    for it in range(2):
        c, d, tc, tv = pkg.synth.make_batch(2, side=256, rank=7, step=it, invalid_frac=0.2)
        rot = np.linalg.qr(np.random.Generator(np.random.PCG64(it)).standard_normal((2, 3, 3)))[0].astype(np.float32)
        batches.append(tuple(torch.from_numpy(a) for a in (c, d, tc, tv, rot)))
    record = trainer.test(1, batches)
    want = json.loads(str(g['test_record']))
    assert set(record) == set(want)
    assert not model.training
    assert record['test_loss'] == pytest.approx(want['test_loss'], rel=1e-3)
    assert record['cam_mean'] == pytest.approx(want['cam_mean'], rel=1e-3)
    for k in ('score_pck', 'score_auc', 'solid', 'close', 'depth', 'jitter', 'switch', 'fail'):
        assert record[k] == pytest.approx(want[k], abs=2e-3), k

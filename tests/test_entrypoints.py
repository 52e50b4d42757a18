"""Upper face of the drop-in boundary: Logger checkpoint formats (log.py:32-40) and the entry-point flows
depth_main.main / main.main (train -> test -> record -> -val_only / -resume) on synthetic loaders."""
import json
import os

import pytest
import torch

THRESH = dict(score=150.0, solid=50.0, close=100.0, rough=150.0, perfect=25.0, good=50.0, jitter=100.0)


def _flags(tmp_path, model='resnet18', extra=()):
    meta = tmp_path / 'metadata.json'
    if not meta.exists():
        meta.write_text(json.dumps(dict(loader=dict(h36m='depth_datasets'), no_depth=dict(h36m=False), thresholds=dict(h36m=THRESH),
                                        root=dict(h36m=str(tmp_path)))))
    return ['-model', model, '-suffix', 'e2e', '-data_name', 'h36m', '-save_path', str(tmp_path / 'runs'), '-criterion', 'SmoothL1',
            '-num_joints', '17', '-side_in', '128', '-batch_size', '2', '-workers', '0', '-synthetic', '2', '-metadata', str(meta)] + list(extra)


def test_logger_formats(pkg, tmp_path):
    args = pkg.opts.parse(_flags(tmp_path, extra=['-save_record']))
    net = torch.nn.Linear(3, 2)
    logger = pkg.log.Logger(args, None)
    assert logger.state == dict(best_auc=0, best_pck=0, best_epoch=0, epoch=0)
    logger.record(1, dict(cam_train_loss=3.0), dict(test_loss=2.0, score_auc=0.2, score_pck=0.3), net)
    logger.record(2, dict(cam_train_loss=2.0), dict(test_loss=1.5, score_auc=0.1, score_pck=0.2), net)       # worse: best stays 1
    root = tmp_path / 'runs' / 'resnet18-e2e'
    ck = torch.load(root / 'model_2.pth')
    assert set(ck) == {'state', 'model'} and set(ck['model']) == {'weight', 'bias'}
    assert ck['state'] == dict(best_auc=0.2, best_pck=0.3, best_epoch=1, epoch=2)
    assert torch.load(root / 'best.pth') == {'best': 1}
    rec = torch.load(root / 'train_record.pth')
    assert rec['cam_train_loss'] == [3.0, 2.0] and rec['score_pck'] == [0.3, 0.2]
    # -resume continues the record and the epoch counter
    args2 = pkg.opts.parse(_flags(tmp_path, extra=['-save_record', '-resume']))
    again = pkg.log.Logger(args2, ck['state'])
    assert again.state['epoch'] == 2 and again.train_record['cam_train_loss'] == [3.0, 2.0]
    with pytest.raises(AssertionError):                       # log.py:18: evaluation runs do not save records
        pkg.log.Logger(pkg.opts.parse(_flags(tmp_path, extra=['-save_record', '-val_only'])), None)


def test_get_loader_follows_metadata(pkg, tmp_path):
    args = pkg.opts.parse(_flags(tmp_path))
    assert pkg.depth_train.get_loader(args).__name__.endswith('.depth_datasets')
    (tmp_path / 'metadata.json').write_text(json.dumps(dict(loader=dict(h36m='datasets'), no_depth=dict(h36m=True),
                                                            thresholds=dict(h36m=THRESH))))
    assert pkg.depth_train.get_loader(args).__name__.endswith('.datasets')


@pytest.mark.gpu
def test_depth_main_train_eval_resume(pkg, tmp_path):
    state = pkg.depth_main.main(_flags(tmp_path, extra=['-n_epochs', '2', '-save_record']))
    assert state['epoch'] == 2 and state['best_epoch'] in (1, 2)
    root = tmp_path / 'runs' / 'resnet18-e2e'
    ck = torch.load(root / 'model_2.pth')
    keys = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'state_keys.json')))['depthnet.resnet18']
    assert {k: list(v.shape) for k, v in ck['model'].items()} == keys['state']     # the reference's names and shapes -> checkpoints interchange
    rec = torch.load(root / 'train_record.pth')
    assert len(rec['cam_train_loss']) == 2 and len(rec['score_auc']) == 2
    val = pkg.depth_main.main(_flags(tmp_path, extra=['-n_epochs', '2', '-val_only']))       # loads model_2.pth (depth_main.py:50-61)
    assert val['test_loss'] == pytest.approx(rec['test_loss'][-1], rel=1e-5)
    more = pkg.depth_main.main(_flags(tmp_path, extra=['-n_epochs', '3', '-save_record', '-resume', '-model_path', str(root / 'model_2.pth')]))
    assert more['epoch'] == 3 and len(torch.load(root / 'train_record.pth')['cam_train_loss']) == 3


@pytest.mark.gpu
def test_legacy_main_flow(pkg, tmp_path):
    (tmp_path / 'metadata.json').write_text(json.dumps(dict(loader=dict(h36m='datasets'), no_depth=dict(h36m=True),
                                                            thresholds=dict(h36m=THRESH))))
    state = pkg.main.main(_flags(tmp_path, extra=['-n_epochs', '1', '-save_record']))
    assert state['epoch'] == 1 and state['best_epoch'] == 1
    root = tmp_path / 'runs' / 'resnet18-e2e'
    assert 'cam_regressor.weight' in torch.load(root / 'model_1.pth')['model']
    val = pkg.main.main(_flags(tmp_path, extra=['-val_only']))                                  # best.pth -> model_1.pth (main.py:38-47)
    assert val['test_loss'] == pytest.approx(torch.load(root / 'train_record.pth')['test_loss'][-1], rel=1e-5)

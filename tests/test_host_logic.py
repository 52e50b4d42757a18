"""Host-side logic that needs no GPU: flag parity with the reference's opts.py, model factories and state-dict
key inventories, the flat parameter layout, the bucket plan, the learning-rate schedule, the loader API and the
world_size-2 gradient exchange over gloo."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import golden_path

BASE = ['-suffix', 't', '-data_name', 'h36m', '-save_path', '/tmp/p3d', '-criterion', 'SmoothL1', '-num_joints', '17', '-side_in', '256']


def parse(pkg, model, extra=()):
    return pkg.opts.parse(['-model', model] + BASE + list(extra))


def test_opts_defaults_match_reference(pkg):
    a = pkg.opts.parse(['-model', 'resnet50', '-suffix', 's', '-data_name', 'ntu', '-save_path', 'x', '-criterion', 'SmoothL1'])
    # reference opts.py:50-76
    assert (a.warmup, a.n_epochs, a.batch_size, a.semi_batch, a.n_cudas, a.workers, a.num_processes) == (1, 20, 64, 16, 2, 2, 6)
    assert (a.side_in, a.stride, a.num_joints, a.depth, a.alpha_span) == (257, 16, 19, 16, 10)
    assert (a.warmup_factor, a.learn_rate, a.learn_decay, a.grad_norm, a.grad_scaling) == (0.2, 5e-5, 0.2, 5.0, 32.0)
    assert (a.momentum, a.weight_decay, a.box_margin, a.depth_range, a.random_zoom, a.loss_div) == (0.9, 4e-5, 0.6, 1000.0, 0.9, 10.0)
    assert not any([a.half_acc, a.do_fusion, a.partial_conv, a.depth_only, a.do_teach, a.colour, a.eraser, a.occluder])
    with pytest.raises(SystemExit):
        pkg.opts.parse(['-model', 'resnet50'])          # the five required flags (opts.py:39-47)


@pytest.mark.parametrize('tag,extra,module', [
    ('depthnet', [], 'depthnet'), ('depthnet_depth_only', ['-depth_only'], 'depthnet'),
    ('fusionnet', ['-do_fusion'], 'fusionnet'), ('partial_depthnet', ['-depth_only', '-partial_conv'], 'partial_depthnet'),
    ('partial_fusionnet', ['-do_fusion', '-partial_conv'], 'partial_fusionnet')])
@pytest.mark.parametrize('model', ['resnet18', 'resnet50'])
def test_state_dict_keys_match_reference(pkg, tag, extra, module, model):
    """Checkpoint interchange (log.py:32-40): same keys, order and shapes as the reference factories."""
    with open(golden_path('state_keys.json')) as f:
        inv = json.load(f)[tag + '.' + model]
    net, _ = pkg.depth_main.create_model(parse(pkg, model, extra))
    assert type(net).__module__.endswith(module)
    got = {k: list(v.shape) for k, v in net.state_dict().items()}
    assert list(got) == list(inv['state'])
    assert got == inv['state']
    assert [n for n, _ in net.named_parameters()] == inv['params']


def test_legacy_resnet_keys(pkg):
    with open(golden_path('state_keys.json')) as f:
        inv = json.load(f)['resnet_joint_extra.resnet50']
    net = pkg.resnet.resnet50(parse(pkg, 'resnet50', ['-joint_space', '-extra_channel']))
    assert {k: list(v.shape) for k, v in net.state_dict().items()} == inv['state']


def test_initialisation_rules(pkg):
    """Init rules of depthnet.py:148-156: kaiming fan_out convs, BN 1/0, regressor left at torch's default."""
    torch.manual_seed(7)
    net, _ = pkg.depth_main.create_model(parse(pkg, 'resnet18'))
    w = net.conv1.weight.detach()
    assert w.shape == (64, 3, 7, 7)
    assert abs(float(w.std()) - (2.0 / (7 * 7 * 64)) ** 0.5) < 0.1 * (2.0 / (7 * 7 * 64)) ** 0.5      # kaiming_normal_(fan_out)
    assert float(net.bn1.weight.min()) == 1.0 and float(net.bn1.bias.abs().max()) == 0.0
    fan_in = 512 * 9
    assert float(net.regressor.weight.detach().abs().max()) <= 1.0 / fan_in ** 0.5 + 1e-6     # torch default init kept (depthnet.py:156)


def test_stage_geometry(pkg):
    g = pkg._trunk.stage_geometry if hasattr(pkg, '_trunk') else __import__('importlib').import_module(pkg.__name__ + '._trunk').stage_geometry
    assert g(16) == ((2, 2, 1), (1, 1, 2))
    assert g(32) == ((2, 2, 2), (1, 1, 1))
    assert g(8) == ((2, 1, 1), (1, 2, 4))
    assert g(4) == ((1, 1, 1), (2, 4, 8))


def test_flat_layout_and_views(pkg):
    offs, total = pkg.optim.plan_layout([10, 3, 16, 1])
    assert offs == [0, 12, 16, 32] and total == 36
    lin = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    before = [p.detach().clone() for p in lin.parameters()]
    opt = pkg.optim.FlatAdam(list(lin.named_parameters()), lr=1e-3, weight_decay=4e-5)
    assert opt.total % 4 == 0 and opt.flat_p.numel() == opt.total
    for p, b, (name, off, n) in zip(lin.parameters(), before, opt.slices()):
        assert torch.equal(p, b)                                            # values preserved
        assert p.data_ptr() == opt.flat_p.data_ptr() + 4 * off              # parameter is a view of the flat buffer
        assert p.grad.data_ptr() == opt.flat_g.data_ptr() + 4 * off
    lin(torch.randn(4, 5)).sum().backward()
    assert float(opt.flat_g.abs().sum()) > 0                                # autograd accumulated straight into the flat buffer
    opt.zero_grad()
    assert float(opt.flat_g.abs().sum()) == 0
    assert len(opt.param_groups) == 2                                       # adapt_learn_rate writes both (depth_train.py:637-638)


def test_bucket_plan_covers_everything_last_first(pkg):
    slices = [('a', 0, 100), ('b', 100, 50), ('c', 152, 1000), ('d', 1152, 8), ('e', 1160, 300)]
    buckets = pkg.dist.plan_buckets(slices, bucket_bytes=4 * 400, total=1460)
    assert buckets[0][1] == 1460 and buckets[-1][0] == 0
    for (s0, e0, _), (s1, e1, _) in zip(buckets, buckets[1:]):
        assert e1 == s0                                                     # contiguous, descending
    assert sorted(i for _, _, m in buckets for i in m) == [0, 1, 2, 3, 4]
    assert buckets[0][2][0] == 4                                            # the last parameter (regressor) goes first


def test_trainer_lr_schedule_and_guards(pkg):
    args = parse(pkg, 'resnet18')
    net, _ = pkg.depth_main.create_model(args)
    tr = pkg.depth_train.Trainer(args, net, pkg.utils.get_info())
    want = {1: 1e-5, 2: 5e-5, 15: 5e-5, 16: 1e-5, 20: 1e-5, 21: 2e-6, 25: 2e-6, 26: 4e-7}      # depth_train.py:621-638
    for epoch, lr in want.items():
        tr.adapt_learn_rate(epoch)
        assert tr.optimizer.param_groups[0]['lr'] == pytest.approx(lr)
        assert tr.optimizer.param_groups[1]['lr'] == pytest.approx(lr)
    assert tr.data_info.key_index == 16 and tr.data_info.short_names[16] == 'pelv'
    with pytest.raises(pkg._lib.P3DError):                                  # and, like fp32, only on the GPU
        pkg.depth_train.Trainer(parse(pkg, 'resnet18', ['-half_acc']), pkg.depth_main.create_model(args)[0], pkg.utils.get_info())
    # no CPU fallback: a forward on CPU tensors must fail loudly, not run on ATen
    with pytest.raises(pkg._lib.P3DError):
        net(torch.zeros(1, 3, 64, 64))


def test_synthetic_loader_contract(pkg):
    """Tuple contract of depth_datasets.py:236-237 / datasets.py:141-146."""
    args = parse(pkg, 'resnet18', ['-synthetic', '2', '-batch_size', '3', '-workers', '0', '-side_in', '64'])
    loader = pkg.depth_datasets.data_loader(args, 'train', pkg.utils.get_info())
    color, depth, cam, val = next(iter(loader))
    assert color.shape == (3, 3, 64, 64) and depth.shape == (3, 1, 64, 64) and cam.shape == (3, 17, 3) and val.shape == (3, 17)
    assert color.dtype == torch.float32 and val.dtype == torch.bool and len(loader) == 2
    assert float((depth == 0).float().mean()) > 0.2                         # ~30 % holes feed the partial-conv mask
    rgb = pkg.datasets.data_loader(args, 'valid', pkg.utils.get_info())
    assert len(next(iter(rgb))) == 4                                        # (color, cam, valid, back_rotate)
    with pytest.raises(FileNotFoundError):                                  # neither -synthetic nor a metadata.json with the dataset root
        pkg.depth_datasets.data_loader(parse(pkg, 'resnet18'), 'train', pkg.utils.get_info())


# ---------------------------------------------------------------------------------------------------
def _ddp_worker(rank, world, port, pkg_name, out_dir):
    import importlib
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    pkg = importlib.import_module(pkg_name)
    r, w, _ = pkg.dist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(0)                                                    # identical weights on both ranks
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 4), torch.nn.Linear(4, 3))
    opt = pkg.optim.FlatAdam(list(net.named_parameters()), lr=1e-3)
    red = pkg.dist.GradReducer(opt, bucket_bytes=64)                        # tiny buckets -> several all-reduces, launched from hooks
    assert len(red.buckets) >= 3
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(8, 6, generator=g)
    opt.zero_grad()
    net(x).pow(2).sum().backward()
    scale = red.finish()
    val = torch.tensor([True, True, False, True]) if rank == 0 else torch.tensor([True, False, False, False])
    div = pkg.dist.global_valid_divisor(val)
    torch.save(dict(flat=opt.flat_g.clone(), scale=scale, div=div, x=x), os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_exchange_world2_gloo(pkg, tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_ddp_worker, args=(2, port, pkg.__name__, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r)) for r in (0, 1))
    assert torch.equal(r0['flat'], r1['flat'])                              # both ranks hold the same reduced buffer
    assert r0['scale'] == 0.5
    assert float(r0['div']) == pytest.approx(3 * 4 / 2)                     # 3 * (3 + 1 valid joints) / world
    # single-process reference: sum of the two ranks' gradients
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 4), torch.nn.Linear(4, 3))
    total = None
    for r in (r0, r1):
        net.zero_grad()
        net(r['x']).pow(2).sum().backward()
        flat = torch.cat([torch.nn.functional.pad(p.grad.reshape(-1), (0, (-p.numel()) % 4)) for p in net.parameters()])
        total = flat if total is None else total + flat
    assert torch.allclose(r0['flat'], total, rtol=1e-5, atol=1e-6)


def test_oracle_crop_warp_properties():
    """The numpy restatement of cameralib.reproject_image_fast's remap: identity, integer shift with zero border, half-pixel mean."""
    from oracle import np_ops as ref
    rng = np.random.default_rng(1)
    img = (rng.random((9, 11, 3)) * 255).astype(np.uint8)
    eye = np.eye(3, dtype=np.float32)
    assert np.array_equal(ref.warp_crop(img, eye, (9, 11)), img.transpose(2, 0, 1).astype(np.float32))
    shift = np.array([[1, 0, 2], [0, 1, -1], [0, 0, 1]], np.float32)          # crop pixel (x, y) reads frame pixel (x + 2, y - 1)
    out = ref.warp_crop(img, shift, (9, 11))
    assert np.array_equal(out[:, 1:, :9], img.transpose(2, 0, 1)[:, :8, 2:].astype(np.float32))
    assert not out[:, 0].any() and not out[:, :, 9:].any()                    # outside the frame: border value 0
    half = np.array([[1, 0, 0.5], [0, 1, 0], [0, 0, 1]], np.float32)
    f = img[..., :1].astype(np.float32)
    got = ref.warp_crop(f, half, (9, 10))
    assert np.allclose(got[0], 0.5 * (f[:, :10, 0] + f[:, 1:11, 0]))
    k = np.array([[100.0, 0, 50], [0, 100.0, 40], [0, 0, 1]])
    assert np.allclose(ref.crop_homography(k, np.eye(3), k, np.eye(3)), np.eye(3), atol=1e-6)


# ---------------------------------------------------------------------------------------------------
def _replica_worker(rank, world, port, pkg_name, out_dir):
    """Ranks start from DIFFERENT random initialisations (no shared seed, as under torchrun); Trainer-style set-up must make them one replica.
    Then a step in which the network runs forward twice before one backward (-semi_teach): per-use "gradient ready" reports must not start a
    bucket's all-reduce while a later use still accumulates into it."""
    import importlib
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    pkg = importlib.import_module(pkg_name)
    pkg.dist.init_from_env(backend='gloo')
    torch.manual_seed(1234 + 77 * rank)                                     # a different model on every rank
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    with torch.no_grad():
        net[1].running_mean.add_(float(rank + 1))
    opt = pkg.optim.FlatAdam(list(net.named_parameters()), lr=1e-3)
    before = opt.flat_p.clone()
    red = pkg.dist.GradReducer(opt, bucket_bytes=32, model=net)
    pkg.dist.broadcast_state(opt, net)
    after, running_mean = opt.flat_p.clone(), net[1].running_mean.clone()
    # two forward passes, one backward; the HIP kernels' in-place gradient writes are mimicked by calling the ready-callback once per use
    g = torch.Generator().manual_seed(500 + rank)
    xa, xb = torch.randn(8, 6, generator=g), torch.randn(8, 6, generator=g)
    opt.zero_grad()
    loss = net(xa).pow(2).sum() + net(xb).pow(2).sum()
    assert red._forwards == 2
    for p in opt.params:                                                   # "first use finished": must NOT launch anything
        p._p3d_grad_ready()
    launched_early = any(red._launched)
    loss.backward()
    for p in opt.params:
        p._p3d_grad_ready()
    scale = red.finish()
    torch.save(dict(before=before, after=after, running_mean=running_mean, flat_g=opt.flat_g.clone(), xa=xa, xb=xb,
                    launched_early=launched_early, scale=scale), os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_replicas_start_identical_and_two_forward_step_reduces_once(pkg, tmp_path):
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_replica_worker, args=(2, port, pkg.__name__, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r)) for r in (0, 1))
    assert not torch.equal(r0['before'], r1['before'])                      # the ranks really started apart ...
    assert torch.equal(r0['after'], r1['after']) and torch.equal(r0['after'], r0['before'])      # ... and continue from rank 0's replica
    assert torch.equal(r0['running_mean'], r1['running_mean'])
    assert not r0['launched_early'] and not r1['launched_early']
    assert torch.equal(r0['flat_g'], r1['flat_g']) and r0['scale'] == 0.5
    # single-process reference: both ranks' two-pass gradients summed, on rank 0's weights
    torch.manual_seed(1234)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    total = None
    for r in (r0, r1):
        net.zero_grad()
        (net(r['xa']).pow(2).sum() + net(r['xb']).pow(2).sum()).backward()
        flat = torch.cat([torch.nn.functional.pad(p.grad.reshape(-1), (0, (-p.numel()) % 4)) for p in net.parameters()])
        total = flat if total is None else total + flat
    assert torch.allclose(r0['flat_g'], total, rtol=1e-5, atol=1e-6)


def test_shards_have_equal_length(pkg, monkeypatch):
    """129 samples on 2 ranks, batch 64: unequal shards would give one rank a third batch and hang the other in the all-reduce."""
    samples = list(range(129))
    lens = []
    for rank in (0, 1):
        monkeypatch.setenv('WORLD_SIZE', '2')
        monkeypatch.setenv('RANK', str(rank))
        part = pkg.depth_datasets.shard(samples, 'train')
        lens.append(len(part))
        assert pkg.depth_datasets.shard(samples, 'valid') == samples          # evaluation is not sharded
    assert lens == [64, 64]
    monkeypatch.setenv('WORLD_SIZE', '1')
    monkeypatch.setenv('RANK', '0')
    assert pkg.depth_datasets.shard(samples, 'train') == samples


def test_rccl_overlap_verdict_fails_closed(pkg, monkeypatch):
    """dist.rccl_overlap_allowed: collectives may run beside the backward pass only with NCCL_ALGO=Ring in this process's environment AND the scanned librccl build
    mapped; anything else -- a launcher's own NCCL_ALGO, no library to verify, another build -- gives (False, reason), and P3D_RCCL_OVERLAP overrides either way."""
    d = pkg.dist
    def verdict(env, mapped=None, size=None):
        for k in ('NCCL_ALGO', 'P3D_RCCL_OVERLAP'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        monkeypatch.setattr(d, '_overlap_verdict', None)
        monkeypatch.setattr(d, '_loaded_rccl_path', lambda: mapped)
        if size is not None:
            monkeypatch.setattr(d.os.path, 'getsize', lambda p: size)
            monkeypatch.setattr(d.os.path, 'exists', lambda p: True)
        return d.rccl_overlap_allowed()
    ok, why = verdict({})
    assert not ok and 'Ring' in why                               # NCCL_ALGO unset: RCCL may pick the tree kernels
    ok, why = verdict({'NCCL_ALGO': 'Tree'})
    assert not ok and 'Tree' in why
    ok, why = verdict({'NCCL_ALGO': 'Ring'})
    assert not ok and 'no librccl' in why                         # nothing mapped to verify
    ok, why = verdict({'NCCL_ALGO': 'ring'}, mapped='/somewhere/librccl.so', size=123)
    assert not ok and 'not the librccl build' in why              # another build: its kernels were never looked at
    assert verdict({'NCCL_ALGO': 'Tree', 'P3D_RCCL_OVERLAP': '1'})[0]
    assert not verdict({'NCCL_ALGO': 'Ring', 'P3D_RCCL_OVERLAP': '0'})[0]
    monkeypatch.setattr(d, '_overlap_verdict', None)

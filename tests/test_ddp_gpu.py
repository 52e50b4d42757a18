"""Data-parallel step on the GPU: two processes (gloo over the one card of the test box; RCCL needs one GPU per rank) run
Trainer.train_step on different shards, and the result must equal the single-process composition of the same pieces:
per-rank gradients with per-rank BN statistics and the GLOBAL valid-joint divisor, summed, scaled by 1/world, clipped, Adam --
composed once from the HIP kernels (exchange logic, to 1e-5) and once from the ORACLE's CPU port (arithmetic: every parameter's
reduced gradient and both ranks' losses against oracle/torch_port.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FLAGS = ['-model', 'resnet18', '-suffix', 't', '-data_name', 'h36m', '-save_path', '/tmp/p3d', '-criterion', 'SmoothL1',
         '-num_joints', '17', '-side_in', '128']


def _make(pkg):
    args = pkg.opts.parse(FLAGS)
    model, _ = pkg.depth_main.create_model(args)
    det = pkg.synth.det_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 0)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in det.items()})
    return args, model.cuda().train()


def _batch(pkg, rank):
    c, d, tc, tv = pkg.synth.make_batch(2, side=128, rank=rank, step=0, invalid_frac=0.3 if rank == 1 else 0.0)
    return torch.from_numpy(c).cuda(), torch.from_numpy(tc).cuda(), torch.from_numpy(tv).cuda()


def _worker(rank, world, port, pkg_name, out_dir):
    import importlib
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                      P3D_DIST_BACKEND='gloo')
    pkg = importlib.import_module(pkg_name)
    pkg.dist.init_from_env()
    args, model = _make(pkg)
    trainer = pkg.depth_train.Trainer(args, model, pkg.utils.get_info(), reducer_bucket_bytes=4 << 20)
    assert trainer.world == 2 and len(trainer.reducer.buckets) > 3
    trainer.verbose = False
    trainer.adapt_learn_rate(1)
    color, cam, val = _batch(pkg, rank)
    loss = trainer.train_step(color, None, cam, val)
    torch.cuda.synchronize()
    torch.save(dict(flat_p=trainer.optimizer.flat_p.cpu(), flat_g=trainer.optimizer.flat_g.cpu(), loss=float(loss), norm=trainer.optimizer.total_norm(0.5)),
               os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_composed_reference(pkg, tmp_path):
    import torch.multiprocessing as mp
    port = 29700 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(2, port, pkg.__name__, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r)) for r in (0, 1))
    assert torch.equal(r0['flat_p'], r1['flat_p'])                       # identical update on every rank, no broadcast needed
    assert r0['norm'] == pytest.approx(r1['norm'], rel=1e-6)

    # single-process composition of the same step
    ops = pkg.ops
    args, model = _make(pkg)
    opt = pkg.optim.FlatAdam(list(model.named_parameters()), args.learn_rate * args.warmup_factor, weight_decay=args.weight_decay)
    batches = [_batch(pkg, r) for r in (0, 1)]
    n_valid = sum(int(b[2].sum()) for b in batches)
    divisor = torch.tensor([3.0 * n_valid / 2], dtype=torch.float32, device='cuda')
    total = torch.zeros_like(opt.flat_g)
    losses = []
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    for color, cam, val in batches:
        model.load_state_dict(state0)                                   # BN running stats: each rank starts from the same buffers
        opt.zero_grad()
        z, _ = model(color)
        relat = ops.softargmax3d(z, 16, 17, 8, 8, 1000.0)
        loss, _ = ops.pose_loss(relat, cam, val, 16, 10.0, 'SmoothL1', count_override=divisor)
        loss.backward()
        total += opt.flat_g
        losses.append(float(loss))
    opt.flat_g.copy_(total)
    assert (total.cpu() - r0['flat_g']).abs().max().item() < 1e-5 * total.abs().max().item()      # the reduced gradient buffer itself
    # The same composition on the ORACLE (oracle/torch_port.py, PyTorch on the CPU): per-rank forward / backward from the same weights with per-rank BatchNorm
    # statistics; the oracle's loss is the local mean over 3 n_r elements, the data-parallel step divides by 3 N / world, so rank r's gradient is the oracle's
    # times world * n_r / N.  This is the check that does not compare the HIP kernels with themselves (depth_main.py:72 spreads the model over GPUs; each replica
    # normalises with its own batch statistics).
    from oracle.torch_port import TorchPort
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    composed, oracle_losses = {}, []
    for r in (0, 1):
        port = TorchPort(pkg.synth.det_state_dict(shapes, 0), family='depthnet', model='resnet18')
        c, d, tc, tv = pkg.synth.make_batch(2, side=128, rank=r, step=0, invalid_frac=0.3 if r == 1 else 0.0)
        want = port.train_step(c, d, tc, tv)
        n_r = int(tv.sum())
        oracle_losses.append(want['loss'] * 3 * n_r / float(divisor))
        for k, g in want['grads'].items():
            composed[k] = composed.get(k, 0) + g * (2.0 * n_r / n_valid)
    assert r0['loss'] == pytest.approx(oracle_losses[0], rel=1e-4) and r1['loss'] == pytest.approx(oracle_losses[1], rel=1e-4)
    worst = 0.0
    for name, prm in model.named_parameters():
        got, ref_g = prm.grad.detach().cpu().numpy(), composed[name]           # prm.grad is a view of opt.flat_g == the reduced buffer
        worst = max(worst, float(np.abs(got - ref_g).max() / max(np.abs(ref_g).max(), 1e-12)))
    assert worst < 2e-3, worst
    opt.clip_and_step(args.grad_norm, grad_scale=0.5)
    torch.cuda.synchronize()
    assert r0['loss'] == pytest.approx(losses[0], rel=1e-5) and r1['loss'] == pytest.approx(losses[1], rel=1e-5)
    assert r0['norm'] == pytest.approx(opt.total_norm(0.5), rel=1e-5)
    diff = (opt.flat_p.cpu() - r0['flat_p']).abs().max().item()
    assert diff < 1e-6, diff

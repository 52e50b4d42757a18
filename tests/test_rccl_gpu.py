"""The RCCL path itself on ONE GPU: a single-rank `nccl` process group (P3D_FORCE_DIST=1) carries every gradient bucket of a training step through librccl --
communicator set-up, the hand-over from the weight-gradient stream, bucket launches from the backward hooks -- and the step must be bit-equal to the same step
without a group (a one-rank sum is the identity).  Runs in a child process: a process group, RCCL's streams and NCCL_ALGO are process-wide state.
Replaces depth_main.py:72 (nn.DataParallel) -> dist.GradReducer; the N > 1 runs are the driver's (bench.py --gpus N)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import importlib, json, os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
pkg = importlib.import_module(%(pkg)r)
FLAGS = ['-model', 'resnet18', '-suffix', 't', '-data_name', 'h36m', '-save_path', '/tmp/p3d', '-criterion', 'SmoothL1', '-num_joints', '17', '-side_in', '128']

def make():
    args = pkg.opts.parse(FLAGS)
    model, _ = pkg.depth_main.create_model(args)
    det = pkg.synth.det_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 0)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in det.items()})
    return args, model.cuda().train()

c, d, tc, tv = pkg.synth.make_batch(4, side=128, rank=0, step=0)
batch = (torch.from_numpy(c).cuda(), None, torch.from_numpy(tc).cuda(), torch.from_numpy(tv).cuda())
out = {}
pkg.dist.init_from_env()
out['backend'] = dist.get_backend()
out['world'] = dist.get_world_size()
out['nccl_algo'] = os.environ.get('NCCL_ALGO')
args, model = make()
trainer = pkg.depth_train.Trainer(args, model, pkg.utils.get_info(), reducer_bucket_bytes=4 << 20)
trainer.verbose = False
trainer.adapt_learn_rate(1)
red = trainer.reducer
out['active'] = bool(red.active)
out['overlap'] = bool(red.overlap)
out['overlap_reason'] = red.overlap_reason
out['buckets'] = len(red.buckets)
calls = []
real = dist.all_reduce
in_finish = [False]
def spy(tensor, *a, **k):
    if tensor.numel() > 1:
        side = pkg.ops._side_stream(tensor.device) if pkg.ops.WGRAD_STREAM else None
        calls.append(dict(numel=tensor.numel(), on_side=bool(side is not None and torch.cuda.current_stream() == side), in_finish=in_finish[0]))
    return real(tensor, *a, **k)
dist.all_reduce = spy
fin = red.finish
def finish():
    in_finish[0] = True
    try:
        return fin()
    finally:
        in_finish[0] = False
red.finish = finish
loss_a = float(trainer.train_step(*batch))
torch.cuda.synchronize()
dist.all_reduce = real
out['calls'] = calls
p_a = trainer.optimizer.flat_p.detach().cpu().clone()
# the same step with no reducer traffic
pkg.dist.FORCE_GROUP = False
args, model2 = make()
trainer2 = pkg.depth_train.Trainer(args, model2, pkg.utils.get_info(), reducer_bucket_bytes=4 << 20)
trainer2.verbose = False
trainer2.adapt_learn_rate(1)
out['active_without'] = bool(trainer2.reducer.active)
loss_b = float(trainer2.train_step(*batch))
torch.cuda.synchronize()
p_b = trainer2.optimizer.flat_p.detach().cpu()
out['loss_equal'] = loss_a == loss_b
out['params_bit_equal'] = bool(torch.equal(p_a, p_b))
out['rccl_mapped'] = pkg.dist._loaded_rccl_path()
dist.destroy_process_group()
print('RESULT ' + json.dumps(out))
'''


def _run(extra_env):
    env = dict(os.environ, P3D_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29800 + os.getpid() % 100),
               RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('NCCL_ALGO', None)
    env.update(extra_env)
    code = CHILD % dict(root=ROOT, pkg='3d-pose-estimation-with-previleged-information_amd')
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('RESULT ')][-1]
    return json.loads(line[7:]), r.stderr


def test_single_rank_rccl_step_is_bit_equal_and_overlapped():
    out, _ = _run({})
    assert out['backend'] == 'nccl' and out['world'] == 1
    assert out['nccl_algo'] == 'Ring'
    assert out['rccl_mapped'] and 'librccl' in out['rccl_mapped']          # librccl is what carried the buckets
    assert out['active'] and not out['active_without']
    assert out['overlap'], out['overlap_reason']                           # the scanned library + ring kernels: collectives may run beside backward
    assert out['buckets'] > 3 and len(out['calls']) == out['buckets']      # every bucket went through dist.all_reduce exactly once
    hooked = [c for c in out['calls'] if not c['in_finish']]
    assert len(hooked) >= out['buckets'] - 1, out['calls']                 # launched from the backward hooks, not saved up for finish()
    assert all(c['on_side'] for c in out['calls']), out['calls']           # issued from the weight-gradient stream
    assert out['loss_equal'] and out['params_bit_equal']


def test_foreign_nccl_algo_gives_up_the_overlap():
    """A launcher that exports NCCL_ALGO=Tree keeps it (setdefault), and the reducer then sends every bucket from finish(), after the backward pass."""
    out, err = _run({'NCCL_ALGO': 'Tree'})
    assert out['nccl_algo'] == 'Tree'
    assert not out['overlap'] and 'Ring' in out['overlap_reason']
    assert 'AFTER the backward pass' in err
    assert all(c['in_finish'] for c in out['calls']) and len(out['calls']) == out['buckets']
    assert out['loss_equal'] and out['params_bit_equal']

#!/usr/bin/env python3
"""Which parameters' reduced gradients differ between the 2-rank step (tests/test_ddp_gpu.py's workers) and the single-process composition?"""
import importlib
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_ddp_gpu as T            # noqa: E402

pkg = importlib.import_module('3d-pose-estimation-with-previleged-information_amd')


def main():
    import torch.multiprocessing as mp
    tmp = tempfile.mkdtemp()
    mp.spawn(T._worker, args=(2, 29811, pkg.__name__, tmp), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp, 'rank0.pt'))
    ops = pkg.ops
    args, model = T._make(pkg)
    opt = pkg.optim.FlatAdam(list(model.named_parameters()), args.learn_rate * args.warmup_factor, weight_decay=args.weight_decay)
    batches = [T._batch(pkg, r) for r in (0, 1)]
    n_valid = sum(int(b[2].sum()) for b in batches)
    divisor = torch.tensor([3.0 * n_valid / 2], dtype=torch.float32, device='cuda')
    total = torch.zeros_like(opt.flat_g)
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    for color, cam, val in batches:
        model.load_state_dict(state0)
        opt.zero_grad()
        z, _ = model(color)
        relat = ops.softargmax3d(z, 16, 17, 8, 8, 1000.0)
        loss, _ = ops.pose_loss(relat, cam, val, 16, 10.0, 'SmoothL1', count_override=divisor)
        loss.backward()
        total += opt.flat_g
    torch.cuda.synchronize()
    ref, got = total.cpu(), r0['flat_g']
    rows = []
    for name, off, n in opt.slices():
        a, b = ref[off:off + n], got[off:off + n]
        d = (a - b).abs().max().item()
        rows.append((d / (a.abs().max().item() + 1e-30), d, a.abs().max().item(), name, bool(torch.equal(a, b))))
    rows.sort(reverse=True)
    print('parameters with bitwise-equal gradients: %d of %d' % (sum(r[4] for r in rows), len(rows)))
    for r in rows[:12]:
        print('rel %.2e  abs %.2e  max|g| %.2e  %s' % r[:4])


if __name__ == '__main__':
    main()

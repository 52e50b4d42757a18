"""FlatAdam: nn.utils.clip_grad_norm_ + optim.Adam(weight_decay) of the reference trainer
(depth_train.py:83,455-456) as two kernel launches over flat HBM buffers.

All parameters are re-homed into ONE contiguous fp32 buffer (each tensor a view at a 16-B aligned
offset), and so are their gradients and the Adam moments.  Consequences:
  * the global L2 norm is one streaming reduction and the update one streaming kernel (28 B/param),
    instead of 161 tensors x ~10 ATen launches for ResNet-50;
  * the clip coefficient is consumed on the device (no .item() between norm and step);
  * the gradient buffer is what the RCCL all-reduce sends, bucket by bucket, with no packing copy.
The reference's two param groups carry identical hyper-parameters (wrap_by_name, depth_train.py:22-25,
weight decay on BN and bias included), so one flat group reproduces it; `param_groups` keeps two dicts
because adapt_learn_rate writes both (depth_train.py:637-638).

Difference from torch.optim.Adam, by construction: the update runs over the WHOLE flat buffer, so a parameter that received no gradient
in a step is still pulled by the weight decay (g = wd * p) and its moments decay, where optim.Adam skips parameters whose .grad is None.
The two agree whenever every registered parameter takes part in every step, which holds for every network / flag combination of the
reference (only `requires_grad` parameters are registered; frozen ones never enter the buffer).  `clip_and_step*` verify that on the first
step of every optimizer (parameters whose gradient range is exactly zero while others are not: a warning), and on every step, as an error, with `P3D_CHECK_GRADS=1`.
"""
import math
import os

import torch

from . import ops

ALIGN = 4   # floats (16 B)


def plan_layout(numels, align=ALIGN):
    """Offsets of each tensor in the flat buffer and the padded total (pure host logic)."""
    offsets, total = [], 0
    for n in numels:
        offsets.append(total)
        total += (n + align - 1) // align * align
    return offsets, total


class FlatAdam:

    def __init__(self, named_params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        named_params = [(n, p) for n, p in named_params if p.requires_grad]
        if not named_params:
            raise ValueError('FlatAdam: no trainable parameter')
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        device = self.params[0].device
        if any(p.device != device or p.dtype != torch.float32 for p in self.params):
            raise ValueError('FlatAdam: parameters must all be fp32 on one device')
        self.offsets, self.total = plan_layout([p.numel() for p in self.params])
        self.flat_p = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                view = self.flat_p[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_g[off:off + p.numel()].view_as(p)
                p._p3d_direct_grad = True        # ops.py backward kernels accumulate straight into this buffer
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.param_groups = [dict(lr=lr), dict(lr=lr)]
        self.step_count = 0
        self._checked = False
        self.norm_sq = torch.zeros(1, dtype=torch.float64, device=device)
        self.dev_state = None            # int32[2] on the device (steps taken, steps skipped) once clip_and_step_dev has been used
        self._dev_scratch = None

    def slices(self):
        """(name, offset, numel) per parameter, in registration order."""
        return [(n, off, p.numel()) for n, p, off in zip(self.names, self.params, self.offsets)]

    def zero_grad(self):
        self.flat_g.zero_()
        for p, off in zip(self.params, self.offsets):        # re-attach if someone dropped a .grad
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                p.grad = self.flat_g[off:off + p.numel()].view_as(p)

    def _check_all_touched(self):
        # always on the first step (a network / flag combination that leaves a registered parameter out of the graph shows up there), afterwards
        # only with P3D_CHECK_GRADS=1: the check reads one flag per parameter back from the device
        if self._checked and not os.environ.get('P3D_CHECK_GRADS'):
            return
        if self.flat_g.is_cuda and torch.cuda.is_current_stream_capturing():
            return                                           # (no read-back inside a graph capture; the first eager step checks)
        self._checked = True
        live = [(n, p) for n, p in zip(self.names, self.params) if p.grad is not None]
        touched = torch.stack([p.grad.ne(0).any() for _, p in live]).cpu().tolist()          # one read-back for all parameters
        missing = [name for (name, _), ok in zip(live, touched) if not ok]
        if not missing or len(missing) == len(live):
            return        # (every gradient exactly zero is a zero loss -- a first batch without a valid joint, an fp16 underflow --, not a parameter out of the graph)
        # The test sees gradient VALUES, not graph membership: a mathematically dead layer looks the same as a parameter that no Function reaches.  torch's Adam
        # (and the reference, depth_train.py:455-456) would carry on either way, so this warns; P3D_CHECK_GRADS=1 makes it an error.
        text = ('FlatAdam: %d of %d parameters have an all-zero gradient this step (%s%s); the flat update still applies weight decay to them'
                % (len(missing), len(live), ', '.join(missing[:4]), ', ...' if len(missing) > 4 else ''))
        if os.environ.get('P3D_CHECK_GRADS'):
            raise RuntimeError(text)
        import warnings
        warnings.warn(text)

    def clip_and_step(self, max_norm, grad_scale=1.0, skip_nonfinite=False):
        """clip_grad_norm_(params, max_norm) followed by Adam.step(); grad_scale (1/world_size) is applied first.
        skip_nonfinite (the -half_acc overflow rule, depth_train.py:431-446): read the norm back and return False WITHOUT stepping
        when any gradient is inf / nan; otherwise returns True."""
        self._check_all_touched()
        self.norm_sq.zero_()
        if (max_norm and max_norm > 0) or skip_nonfinite:
            ops.l2norm_sq_accum(self.flat_g, self.norm_sq)
        if skip_nonfinite and not math.isfinite(float(self.norm_sq.item())):
            return False
        self.step_count += 1
        ops.weights_changed()
        ops.adam_step(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.param_groups[0]['lr'], self.betas[0],
                      self.betas[1], self.eps, self.weight_decay, self.step_count, max_norm or 0.0,
                      self.norm_sq if (max_norm and max_norm > 0) else None, grad_scale)
        from . import ops_block
        ops_block.rebuild_images_early(self.flat_p.device)
        return True

    def clip_and_step_dev(self, max_norm, grad_scale=1.0, skip_nonfinite=True):
        """clip + Adam with the step counter and the overflow-skip decision on the device (-half_acc, depth_train.py:431-446):
        nothing is read back, so the host keeps running ahead of the GPU.  `steps_taken()` / `steps_skipped()` synchronise."""
        if self.dev_state is None:
            self.dev_state = torch.tensor([self.step_count, 0], dtype=torch.int32, device=self.flat_p.device)
            self._dev_scratch = torch.zeros(4, dtype=torch.float32, device=self.flat_p.device)
        self._check_all_touched()
        self.norm_sq.zero_()
        ops.weights_changed()
        ops.l2norm_sq_accum(self.flat_g, self.norm_sq)
        ops.adam_step_dev(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.param_groups[0]['lr'], self.betas[0], self.betas[1],
                          self.eps, self.weight_decay, self.dev_state, max_norm or 0.0, self.norm_sq, grad_scale, skip_nonfinite, self._dev_scratch)

    def steps_taken(self):
        if self.dev_state is not None:
            self.step_count = int(self.dev_state[0].item())
        return self.step_count

    def steps_skipped(self):
        return int(self.dev_state[1].item()) if self.dev_state is not None else 0

    def step(self):
        self.clip_and_step(0.0)

    def total_norm(self, grad_scale=1.0):
        """Host value of the last global gradient norm (synchronises; for logging/tests only)."""
        return float(self.norm_sq.item()) ** 0.5 * grad_scale

    def state_dict(self):
        return dict(step=self.steps_taken(), exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, names=self.names, offsets=self.offsets)

    def load_state_dict(self, state):
        self.step_count = int(state['step'])
        self.dev_state = None
        self.exp_avg.copy_(state['exp_avg'])
        self.exp_avg_sq.copy_(state['exp_avg_sq'])

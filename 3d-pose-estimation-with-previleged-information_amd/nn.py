"""Layer modules with torch.nn's constructor signatures, parameter names and initialisers, whose
forward runs the hand-written HIP kernels (ops.py) instead of ATen.

They subclass the torch.nn classes only to inherit parameter/buffer registration, state_dict
layout and `isinstance` behaviour (the reference's init loops and freeze_batchnorm test
`isinstance(m, nn.Conv2d)` / `nn.BatchNorm2d`: depthnet.py:148-154,158-161).  No ATen compute
kernel is reachable from their forward.
"""
import torch
import torch.nn as nn

from . import ops, ops_half


def _one(v):
    if isinstance(v, (tuple, list)):
        if len(v) != 2 or v[0] != v[1]:
            raise ops.P3DError('only square kernels / symmetric stride, padding, dilation are supported (got %r)' % (v,))
        return int(v[0])
    return int(v)


class Conv2d(nn.Conv2d):
    """nn.Conv2d (groups = 1, zero padding)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.groups != 1 or self.padding_mode != 'zeros' or isinstance(self.padding, str):
            raise ops.P3DError('Conv2d: groups != 1 / non-zero padding modes are not on the hot path')

    def forward(self, x, join_put=None, join_take=None):
        if x.dtype == torch.float16:                        # -half_acc: NHWC fp16 kernels (ops_half.py)
            return ops_half.conv2d(x, self, _one(self.stride), _one(self.padding), _one(self.dilation), join_put, join_take)
        if join_put is None and join_take is None and torch.is_grad_enabled() and self.weight.requires_grad:
            from . import ops_block
            if not x.requires_grad and ops_block.stem_takes_x3(self, x):      # the 7x7 stride-2 stem: restated over a space-to-depth image (csrc/p3d_fx.hip)
                return ops_block.stem_conv(self, x)
            if ops_block.conv_takes_images(self, x):        # multi-tap convolutions outside a residual block (the regressor): operands as pre-split images
                return ops_block.conv2d_images(self, x)
        return ops.conv2d(x, self.weight, self.bias, _one(self.stride), _one(self.padding), _one(self.dilation),
                          join_put=join_put, join_take=join_take)


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d; forward(x, res=None, relu=False) fuses the residual add and the ReLU that the
    reference applies right after it (depthnet.py:42-56,98-116)."""

    def forward(self, x, res=None, relu=False, res_join=None):
        if not (self.affine and self.track_running_stats):
            raise ops.P3DError('BatchNorm2d: only affine=True, track_running_stats=True is on the hot path')
        training = self.training
        momentum = 0.1 if self.momentum is None else self.momentum
        if training:
            if getattr(self, '_ticked', False):       # already counted by the network's pre-hook (_trunk._tick_batchnorm)
                self._ticked = False
            else:
                self.num_batches_tracked.add_(1)
        if x.dtype == torch.float16:
            return ops_half.batch_norm_act(x, self.weight, self.bias, self.running_mean, self.running_var, res, relu, training, momentum, self.eps,
                                           res_join)
        return ops.batch_norm_act(x, self.weight, self.bias, self.running_mean, self.running_var, res, relu, training,
                                  momentum, self.eps, res_join)


class MaxPool2d(nn.MaxPool2d):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) -- the only geometry the reference uses."""

    def forward(self, x):
        if (_one(self.kernel_size), _one(self.stride), _one(self.padding), _one(self.dilation)) != (3, 2, 1, 1) or self.ceil_mode:
            raise ops.P3DError('MaxPool2d: only kernel 3, stride 2, padding 1 is on the hot path')
        return ops_half.maxpool3x3s2(x) if x.dtype == torch.float16 else ops.maxpool3x3s2(x)


class Sequential(nn.Sequential):
    pass

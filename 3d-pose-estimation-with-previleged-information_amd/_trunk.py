"""Shared building blocks of the four network families (depthnet / resnet / fusionnet /
partial_depthnet).  The reference repeats these classes in every file (depthnet.py:10-116,
resnet.py:21-119, fusionnet.py:21-127, partial_depthnet.py:11-157); here they are written once,
on the fused HIP layers, and the family modules only re-export them under the reference's names.

Module attribute names (conv1/bn1/.../downsample.0/.1, layerN.i) reproduce the reference's
state_dict keys so checkpoints interchange (log.py:32-40).
"""
import numpy as np
import torch
import torch.nn as nn

import os

from . import ops, ops_block, ops_half
from .nn import BatchNorm2d, Conv2d, MaxPool2d, Sequential
from .partial_conv import PartialConv


# P3D_BLOCKS=0: every layer as its own autograd node (ops.py), BatchNorm as stand-alone passes -- the round-1 structure, kept as the general path
FUSED_BLOCKS = os.environ.get('P3D_BLOCKS', '1') != '0'


def stage_geometry(stride):
    """(stride2, stride3, stride4), (dilate2, dilate3, dilate4) from the network stride (depthnet.py:130-136)."""
    lg = float(np.log2(stride))
    s2 = int(min(max(lg, 2), 3) - 1)
    s3 = int(min(max(lg, 3), 4) - 2)
    s4 = int(min(max(lg, 4), 5) - 3)
    return (s2, s3, s4), (3 - s2, (3 - s2) * (3 - s3), (3 - s2) * (3 - s3) * (3 - s4))


class _ResidualBlock(nn.Module):
    """conv-bn-relu chain + identity/downsample shortcut; the last BN kernel also adds the shortcut and
    applies the closing ReLU (one pass over the tensor instead of three)."""
    expansion = 1
    kind = 'basic'

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, skip_relu=False, partial=False):
        super().__init__()
        conv = PartialConv if partial else Conv2d
        if self.kind == 'basic':
            self.conv1 = conv(inplanes, planes, kernel_size=3, stride=stride, dilation=dilation, padding=dilation, bias=False)
            self.bn1 = BatchNorm2d(planes)
            self.conv2 = conv(planes, planes, kernel_size=3, padding=1, bias=False)
            self.bn2 = BatchNorm2d(planes)
            self._chain = (('conv1', 'bn1'), ('conv2', 'bn2'))
        else:
            self.conv1 = conv(inplanes, planes, kernel_size=1, bias=False)
            self.bn1 = BatchNorm2d(planes)
            self.conv2 = conv(planes, planes, kernel_size=3, stride=stride, padding=dilation, dilation=dilation, bias=False)
            self.bn2 = BatchNorm2d(planes)
            self.conv3 = conv(planes, planes * 4, kernel_size=1, bias=False)
            self.bn3 = BatchNorm2d(planes * 4)
            self._chain = (('conv1', 'bn1'), ('conv2', 'bn2'), ('conv3', 'bn3'))
        self.downsample = downsample
        self.stride = stride
        self.skip_relu = skip_relu
        self.partial = partial

    def _shortcut(self, x, join):
        """Returns (res, res_join): with a downsample branch its conv hands its input gradient to `join`; an identity shortcut lets
        the closing BN hand over the residual gradient instead."""
        if self.downsample is None:
            return x, join
        return self.downsample[1](self.downsample[0](x, join_put=join)), None

    def forward(self, x):
        if self.partial:                      # partial_depthnet.py:44-46: blocks receive an (x, veil) tuple
            return self.forward_partial(*x)
        # the block input fans out to conv1 and the shortcut: join the two input gradients inside conv1's dgrad kernel (ops.GradJoin)
        if ops.can_fuse_eval(x, self.conv1, self.bn1):
            return self._forward_inference(x)
        if FUSED_BLOCKS and ops_block.usable(self, x):          # training: the whole block is one C call per direction, BatchNorm inside the convolutions
            return ops_block.residual_block(self, x)
        if FUSED_BLOCKS and x.dtype == torch.float16 and ops_half.block_usable(self, x):      # -half_acc: one C call per block and direction over the fp16 kernels
            return ops_half.residual_block(self, x)
        join = ops.GradJoin() if (torch.is_grad_enabled() and x.requires_grad) else None
        out = x
        last = len(self._chain) - 1
        for i, (cname, bname) in enumerate(self._chain):
            out = getattr(self, cname)(out, join_take=join) if i == 0 else getattr(self, cname)(out)
            if i < last:
                out = getattr(self, bname)(out, relu=True)
            else:
                res, res_join = self._shortcut(x, join)
                out = getattr(self, bname)(out, res=res, relu=not self.skip_relu, res_join=res_join)
        return out

    def _forward_inference(self, x):
        """model.eval() under no_grad: every conv + BN (+ shortcut add + ReLU) pair of the block is one fused kernel."""
        res = x if self.downsample is None else ops.conv_bn_eval(x, self.downsample[0], self.downsample[1])
        out = x
        last = len(self._chain) - 1
        for i, (cname, bname) in enumerate(self._chain):
            conv, bn = getattr(self, cname), getattr(self, bname)
            out = ops.conv_bn_eval(out, conv, bn, relu=True) if i < last else ops.conv_bn_eval(out, conv, bn, res=res, relu=not self.skip_relu)
        return out

    def forward_partial(self, x, veil):
        if FUSED_BLOCKS and ops_block.usable(self, x, veil):          # training: the masked block as one C call per direction, like the dense ones
            return ops_block.residual_block(self, x, veil)
        join = ops.GradJoin() if (torch.is_grad_enabled() and x.requires_grad) else None
        out = x
        last = len(self._chain) - 1
        for i, (cname, bname) in enumerate(self._chain):
            out, veil = getattr(self, cname)(out, veil, join_take=join) if i == 0 else getattr(self, cname)(out, veil)
            if i < last:
                out = getattr(self, bname)(out, relu=True)
            else:
                res, res_join = self._shortcut(x, join)                             # shortcut stays dense (partial_depthnet.py:70-75)
                out = getattr(self, bname)(out, res=res, relu=True, res_join=res_join)
        return out, veil


class BasicBlock(_ResidualBlock):
    expansion = 1
    kind = 'basic'


class Bottleneck(_ResidualBlock):
    expansion = 4
    kind = 'bottleneck'


def kaiming_fan_out_(module):
    """Init loop of depthnet.py:148-154 / partial_depthnet.py:187-193."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


def normal_fan_out_(module):
    """Init loop of resnet.py:151-158 / fusionnet.py:186-193."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
            m.weight.data.normal_(0, (2.0 / n) ** 0.5)
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()


def _tick_batchnorm(module, inputs):
    """Forward pre-hook of the whole network: one multi-tensor add bumps `num_batches_tracked` of every BatchNorm2d in training
    mode (53 one-element launches per ResNet-50 step otherwise); each layer then skips its own increment for this call."""
    cache = module.__dict__.get('_bn_layers')
    if cache is None:                                   # (the module tree of these networks is fixed after construction)
        cache = module.__dict__['_bn_layers'] = [m for m in module.modules() if isinstance(m, BatchNorm2d)]
    layers = [m for m in cache if m.training and m.num_batches_tracked.is_cuda]
    if layers:
        torch._foreach_add_([m.num_batches_tracked for m in layers], 1)
        for m in layers:
            m._ticked = True


class TrunkBase(nn.Module):
    """Holds `inplanes` bookkeeping and the stage factory shared by every family."""

    def __init__(self):
        super().__init__()
        self.register_forward_pre_hook(_tick_batchnorm)

    # -half_acc (depth_train.py:73-83): the Trainer sets `_p3d_half`; the network then runs on NHWC fp16 activations between
    # these two conversions, parameters stay fp32 masters with fp16 weight images beside them (ops_half.refresh_weights).
    _p3d_half = False

    def _half_in(self, x):
        from . import ops_half
        return ops_half.to_half_nhwc(x, ops_half.pad8(x.shape[1])) if self._p3d_half else x

    def _half_out(self, *tensors):
        from . import ops_half
        out = tuple(ops_half.to_float(t) if (t is not None and t.dtype == torch.float16) else t for t in tensors)
        return out if len(out) > 1 else out[0]

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1, skip_relu=False, partial=False):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = Sequential(
                Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                BatchNorm2d(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride, dilation, downsample, partial=partial)]
        self.inplanes = planes * block.expansion
        for i in range(1, blocks):
            layers.append(block(self.inplanes, planes, skip_relu=(skip_relu and i == blocks - 1), partial=partial))
        return Sequential(*layers)

    def freeze_batchnorm(self):
        """depthnet.py:158-161"""
        for module in self.modules():
            if isinstance(module, nn.BatchNorm2d):
                module.eval()


def stem_tail(bn, pool, c):
    """maxpool(relu(bn(c))) behind a stem convolution that was called separately (the partial families' PartialConv stems return (c, veil)): in training, fp32,
    one node that never writes the BatchNorm output (ops.stem_tail), otherwise the two layers."""
    if ops.stem_tail_usable(c, bn, pool):
        return ops.stem_tail(c, bn)
    return pool(bn(c, relu=True))


def stem(conv, bn, pool, x):
    """conv -> BN+ReLU (one kernel) -> maxpool; at inference conv + BN + ReLU are one kernel."""
    if ops.can_fuse_eval(x, conv, bn):
        return pool(ops.conv_bn_eval(x, conv, bn, relu=True))
    c = conv(x)
    if ops.stem_tail_usable(c, bn, pool):              # training, fp32: BatchNorm + ReLU + max pool as one node that never writes the BatchNorm output
        return ops.stem_tail(c, bn)
    return pool(bn(c, relu=True))

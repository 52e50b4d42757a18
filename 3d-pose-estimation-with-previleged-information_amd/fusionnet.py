"""Two-stream RGB + depth fusion network (reference fusionnet.py:130-305) on the HIP layers.

RGB: conv1/bn1 -> layer1 -> layer2;  depth: conv2/bn2 -> layer5 -> layer6;  Fusion = 1x1 conv over the
channel concat + BN + ReLU; then layer3, layer4, regressor.  The concat is never materialised: the 1x1
conv reads the two streams through two input-channel windows of its weight (ops.conv_cat1x1).
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._trunk import BasicBlock, Bottleneck, TrunkBase, normal_fan_out_, stage_geometry, stem
from .nn import BatchNorm2d, Conv2d, MaxPool2d

__all__ = ['BasicBlock', 'Bottleneck', 'Fusion', 'ResNet', 'resnet18', 'resnet50']


class Fusion(nn.Module):

    def __init__(self, inplanes):
        super().__init__()
        self.conv = Conv2d(inplanes * 2, inplanes, kernel_size=1, bias=False)
        self.bn = BatchNorm2d(inplanes)

    def forward(self, x, y):
        if x.dtype == torch.float16:                      # -half_acc: NHWC concat kernel + the ordinary fp16 1x1 convolution
            from . import ops_half
            return self.bn(self.conv(ops_half.concat(x, y)), relu=True)
        return self.bn(ops.conv_cat1x1(x, y, self.conv.weight), relu=True)     # fusionnet.py:138-140 (two channel windows: not folded at inference)


class ResNet(TrunkBase):

    def __init__(self, block, layers, args):
        assert args.stride in [4, 8, 16, 32]
        super().__init__()
        self.early_dist = args.early_dist
        self.skip_relu = args.skip_relu
        (s2, s3, s4), (d2, d3, d4) = stage_geometry(args.stride)
        self.conv1 = Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.conv2 = Conv2d(1, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.bn2 = BatchNorm2d(64)
        self.maxpool = MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.inplanes = 64
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=s2, dilation=d2)
        self.fusion = Fusion(self.inplanes)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=s3, dilation=d3, skip_relu=args.skip_relu)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=s4, dilation=d4, skip_relu=args.skip_relu)
        self.inplanes = 64
        self.layer5 = self._make_layer(block, 64, layers[0])
        self.layer6 = self._make_layer(block, 128, layers[1], stride=s2, dilation=d2)
        normal_fan_out_(self)
        self.regressor = Conv2d(512 * block.expansion, args.depth * args.num_joints, 3, padding=1)

    def forward(self, x, y):
        x = stem(self.conv1, self.bn1, self.maxpool, self._half_in(x))
        y = stem(self.conv2, self.bn2, self.maxpool, self._half_in(y))
        x = self.layer2(self.layer1(x))
        y = self.layer6(self.layer5(y))
        x = self.fusion(x, y)
        m = self.layer3(x)
        n = self.layer4(ops.relu(m) if self.skip_relu else m)
        z = self.regressor(ops.relu(n) if self.skip_relu else n)
        return self._half_out(z, m if self.early_dist else n)


def manual_update(model_dict, toy_dict):
    """Seed the depth branch from an RGB pre-train (fusionnet.py:243-262): bn2<-bn1, layer5<-layer1,
    layer6<-layer2, conv2.weight <- first input channel of conv1.weight."""
    manual_keys = set()
    for key in model_dict.keys():
        for dst, src in (('bn2', 'bn1'), ('layer5', 'layer1'), ('layer6', 'layer2')):
            if key.startswith(dst) and key.replace(dst, src) in toy_dict:
                model_dict[key] = toy_dict[key.replace(dst, src)].clone()
                manual_keys.add(key)
    model_dict['conv2.weight'] = toy_dict['conv1.weight'][:, :1].clone()
    manual_keys.add('conv2.weight')
    return manual_keys


def build_resnet(block, layers, args, pretrain):
    model = ResNet(block, layers, args)
    if not pretrain:
        return model
    model_dict = model.state_dict()
    toy_dict = torch.load(args.host_path, map_location='cpu')['model'] if args.depth_host else torch.load(args.model_path, map_location='cpu')
    manual_keys = manual_update(model_dict, toy_dict)
    toy_dict = torch.load(args.model_path, map_location='cpu')
    untended = set(model_dict.keys()).difference(set(toy_dict.keys())).difference(manual_keys)
    untended = [key for key in untended if not key.endswith('num_batches_tracked')]
    assert np.all([key.startswith('fusion') or key.startswith('regressor') for key in untended])   # fusionnet.py:279-285
    for key in list(toy_dict.keys()):
        if key not in model_dict:
            print('toy key [', key, '] discarded')
            del toy_dict[key]
    model_dict.update(toy_dict)
    model.load_state_dict(model_dict)
    return model


def resnet18(args, pretrain):
    return build_resnet(BasicBlock, [2, 2, 2, 2], args, pretrain)


def resnet50(args, pretrain):
    return build_resnet(Bottleneck, [3, 4, 6, 3], args, pretrain)

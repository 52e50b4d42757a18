"""Two-stream fusion network whose depth stream uses partial convolutions (reference partial_fusionnet.py:184-275)
on the HIP layers.

The reference file does not run as shipped: its `conv1` (RGB stem) is built as a PartialConv but called with one
argument and its `conv2` (depth stem) is a plain nn.Conv2d called with (y, veil) (partial_fusionnet.py:202-203 vs
:251,257).  This module implements the wiring its forward() spells out: a dense RGB stem conv1/bn1 -> layer1 -> layer2,
and a partial depth stream conv2(y, veil)/bn2 -> layer5 -> layer6 with veil = (y != 0) max-pooled with the features
(:255-269), then Fusion -> layer3 -> layer4 -> regressor (:271-275).  Parameter names and shapes equal the reference's
(PartialConv adds no parameters), so checkpoints interchange.
"""
import numpy as np
import torch

from . import ops
from ._trunk import BasicBlock, Bottleneck, TrunkBase, normal_fan_out_, stage_geometry, stem, stem_tail
from .fusionnet import Fusion, manual_update
from .nn import BatchNorm2d, Conv2d, MaxPool2d
from .partial_conv import PartialConv

__all__ = ['BasicBlock', 'Bottleneck', 'Fusion', 'ResNet', 'resnet18', 'resnet50']


class ResNet(TrunkBase):

    def __init__(self, block, layers, args):
        assert args.stride in [4, 8, 16, 32]
        super().__init__()
        (s2, s3, s4), (d2, d3, d4) = stage_geometry(args.stride)
        self.conv1 = Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.conv2 = PartialConv(1, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.bn2 = BatchNorm2d(64)
        self.maxpool = MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.inplanes = 64
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=s2, dilation=d2)
        self.fusion = Fusion(self.inplanes)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=s3, dilation=d3)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=s4, dilation=d4)
        self.inplanes = 64
        self.layer5 = self._make_layer(block, 64, layers[0], partial=True)
        self.layer6 = self._make_layer(block, 128, layers[1], stride=s2, dilation=d2, partial=True)
        normal_fan_out_(self)                                                    # partial_fusionnet.py:223-230
        self.regressor = Conv2d(512 * block.expansion, args.depth * args.num_joints, 3, padding=1)

    def forward(self, x, y):
        x = stem(self.conv1, self.bn1, self.maxpool, self._half_in(x))
        with torch.no_grad():
            veil = ops.nonzero_mask(y)                                           # partial_fusionnet.py:255
        y, veil = self.conv2(self._half_in(y), veil)
        y = stem_tail(self.bn2, self.maxpool, y)
        with torch.no_grad():
            veil = self.maxpool(veil)
        x = self.layer2(self.layer1(x))
        y, veil = self.layer5((y, veil))
        y, veil = self.layer6((y, veil))
        x = self.fusion(x, y)
        x = self.layer4(self.layer3(x))
        z = self.regressor(x)
        return self._half_out(z, x)


def build_resnet(block, layers, args, pretrain):
    model = ResNet(block, layers, args)
    if not pretrain:
        return model
    model_dict = model.state_dict()                                              # partial_fusionnet.py:299-329
    toy_dict = torch.load(args.host_path, map_location='cpu')['model'] if args.depth_host else torch.load(args.model_path, map_location='cpu')
    manual_keys = manual_update(model_dict, toy_dict)
    toy_dict = torch.load(args.model_path, map_location='cpu')
    untended = set(model_dict.keys()).difference(set(toy_dict.keys())).difference(manual_keys)
    untended = [key for key in untended if not key.endswith('num_batches_tracked')]
    assert np.all([key.startswith('fusion') or key.startswith('regressor') for key in untended])
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

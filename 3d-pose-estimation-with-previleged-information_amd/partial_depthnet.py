"""Depth network whose stem, layer1 and layer2 use partial convolutions (reference
partial_depthnet.py:160-265) on the HIP layers.

The validity mask ("veil") starts as (x != 0), is max-pooled with the features, and threads through the
partial blocks as an (x, veil) tuple; layer3/4 and every shortcut are dense (partial_depthnet.py:213-229).
"""
import torch

from . import ops
from ._trunk import BasicBlock, Bottleneck, TrunkBase, kaiming_fan_out_, stage_geometry, stem_tail
from .nn import BatchNorm2d, Conv2d, MaxPool2d
from .partial_conv import PartialConv

__all__ = ['BasicBlock', 'Bottleneck', 'ResNet', 'resnet18', 'resnet50']


class ResNet(TrunkBase):

    def __init__(self, block, layers, args):
        assert args.depth_only                                                   # partial_depthnet.py:164
        assert args.stride in [4, 8, 16, 32]
        super().__init__()
        (s2, s3, s4), (d2, d3, d4) = stage_geometry(args.stride)
        self.conv1 = PartialConv(1, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.maxpool = MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.inplanes = 64
        self.layer1 = self._make_layer(block, 64, layers[0], partial=True)
        self.layer2 = self._make_layer(block, 128, layers[1], stride=s2, dilation=d2, partial=True)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=s3, dilation=d3)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=s4, dilation=d4)
        kaiming_fan_out_(self)
        self.regressor = Conv2d(512 * block.expansion, args.depth * args.num_joints, 3, padding=1)

    def forward(self, x):
        with torch.no_grad():
            veil = ops.nonzero_mask(x)                       # fp32 [B,1,H,W] in both precisions
        x, veil = self.conv1(self._half_in(x), veil)
        x = stem_tail(self.bn1, self.maxpool, x)
        with torch.no_grad():
            veil = self.maxpool(veil)
        x, veil = self.layer1((x, veil))
        x, veil = self.layer2((x, veil))
        x = self.layer3(x)
        x = self.layer4(x)
        z = self.regressor(x)
        return self._half_out(z, x)


def build_resnet(block, layers, args, pretrain):
    model = ResNet(block, layers, args)
    if not pretrain:
        return model
    source = torch.load(args.model_path, map_location='cpu')                   # partial_depthnet.py:233-254
    state = model.state_dict()
    source['conv1.weight'] = source['conv1.weight'][:, :1].clone()
    for key in list(source.keys()):
        if key not in state:
            print('key [', key, '] deleted')
            del source[key]
    untended = [key for key in set(state.keys()).difference(set(source.keys())) if not key.endswith('num_batches_tracked')]
    print('keys untended:', untended)
    state.update(source)
    model.load_state_dict(state)
    return model


def resnet18(args, pretrain):
    return build_resnet(BasicBlock, [2, 2, 2, 2], args, pretrain)


def resnet50(args, pretrain):
    return build_resnet(Bottleneck, [3, 4, 6, 3], args, pretrain)

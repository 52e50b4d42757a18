"""Single-stream depth/RGB pose network (reference depthnet.py:119-237) on the HIP layers.

`resnet18(args, pretrain)` / `resnet50(args, pretrain)` return a module whose forward(x) gives
(z, feat): z [B, depth*num_joints, S/stride, S/stride] from the 3x3 `regressor` conv (with bias) and
feat = layer3 output if args.early_dist else layer4 output (depthnet.py:188-200).
"""
import torch

from . import ops
from ._trunk import BasicBlock, Bottleneck, TrunkBase, kaiming_fan_out_, stage_geometry, stem
from .nn import BatchNorm2d, Conv2d, MaxPool2d

__all__ = ['BasicBlock', 'Bottleneck', 'ResNet', 'resnet18', 'resnet50']


class ResNet(TrunkBase):

    def __init__(self, block, layers, args):
        assert args.stride in [4, 8, 16, 32]                                    # depthnet.py:123
        super().__init__()
        self.early_dist = args.early_dist
        self.skip_relu = args.skip_relu
        (s2, s3, s4), (d2, d3, d4) = stage_geometry(args.stride)

        self.conv1 = Conv2d(1 if args.depth_only else 3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.maxpool = MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.inplanes = 64
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=s2, dilation=d2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=s3, dilation=d3, skip_relu=args.skip_relu)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=s4, dilation=d4, skip_relu=args.skip_relu)
        kaiming_fan_out_(self)
        # created after the init loop, so it keeps torch's default initialisation (depthnet.py:156)
        self.regressor = Conv2d(512 * block.expansion, args.depth * args.num_joints, 3, padding=1)

    def forward(self, x):
        x = stem(self.conv1, self.bn1, self.maxpool, self._half_in(x))
        x = self.layer1(x)
        x = self.layer2(x)
        m = self.layer3(x)
        n = self.layer4(ops.relu(m) if self.skip_relu else m)
        z = self.regressor(ops.relu(n) if self.skip_relu else n)
        return self._half_out(z, m if self.early_dist else n)


def build_resnet(block, layers, args, pretrain):
    model = ResNet(block, layers, args)
    if not pretrain:
        return model
    # depthnet.py:204-227: adapt an ImageNet / depth-host checkpoint to this stem, drop foreign keys
    source = torch.load(args.host_path, map_location='cpu')['model'] if args.depth_host else torch.load(args.model_path, map_location='cpu')
    state = model.state_dict()
    if args.depth_only:
        source['conv1.weight'] = source['conv1.weight'][:, :1].clone()
    if args.depth_host:
        source['conv1.weight'] = (source['conv1.weight'] / 3).repeat(1, 3, 1, 1)
    for key in list(source.keys()):
        if key not in state:
            print('key [', key, '] deleted')
            del source[key]
    state.update(source)
    model.load_state_dict(state)
    return model


def resnet18(args, pretrain):
    return build_resnet(BasicBlock, [2, 2, 2, 2], args, pretrain)


def resnet50(args, pretrain):
    return build_resnet(Bottleneck, [3, 4, 6, 3], args, pretrain)

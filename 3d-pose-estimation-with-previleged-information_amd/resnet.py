"""Legacy RGB pose network of main.py (reference resnet.py:122-262) on the HIP layers.

forward(x) returns z_cam, or (z_cam, z_mat) when args.joint_space adds the J-channel image-space head
(resnet.py:196-210).  The stem takes 4 channels under args.extra_channel (resnet.py:142).
"""
import torch

from ._trunk import BasicBlock, Bottleneck, TrunkBase, normal_fan_out_, stage_geometry, stem
from .nn import BatchNorm2d, Conv2d, MaxPool2d

__all__ = ['BasicBlock', 'Bottleneck', 'ResNet', 'resnet18', 'resnet50']


class ResNet(TrunkBase):

    def __init__(self, block, layers, args):
        assert args.stride in [16, 32]                                           # resnet.py:126
        super().__init__()
        self.inplanes = 64
        (s2, s3, s4), (d2, d3, d4) = stage_geometry(args.stride)
        self.conv1 = Conv2d(4 if args.extra_channel else 3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.maxpool = MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=s2, dilation=d2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=s3, dilation=d3)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=s4, dilation=d4)
        normal_fan_out_(self)
        self.cam_regressor = Conv2d(512 * block.expansion, args.depth * args.num_joints, kernel_size=3, padding=1)
        self.mat_regressor = Conv2d(512 * block.expansion, args.num_joints, kernel_size=3, padding=1) if args.joint_space else None

    def forward(self, x):
        x = stem(self.conv1, self.bn1, self.maxpool, self._half_in(x))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        if self.mat_regressor is not None:
            return self._half_out(self.cam_regressor(x), self.mat_regressor(x))
        return self._half_out(self.cam_regressor(x))


def _build(block, layers, args):
    model = ResNet(block, layers, args)
    if not args.pretrain:
        return model
    source = torch.load(args.model_path, map_location='cpu')                   # resnet.py:214-232
    state = model.state_dict()
    if args.extra_channel:
        widened = state['conv1.weight'].clone()
        widened[:, :3] = source['conv1.weight']
        source['conv1.weight'] = widened
    for key in list(source.keys()):
        if key not in state:
            print('key [', key, '] deleted')
            del source[key]
    state.update(source)
    model.load_state_dict(state)
    return model


def resnet18(args):
    return _build(BasicBlock, [2, 2, 2, 2], args)


def resnet50(args):
    return _build(Bottleneck, [3, 4, 6, 3], args)

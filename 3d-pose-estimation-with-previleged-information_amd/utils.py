"""Pose-head helpers with the reference's names (utils.py:146-194), on the fused HIP head kernel.

The reference materialises a [B,J,H,W,D] softmax volume in to_heatmap and reduces it three times in
decode.  Every caller only ever feeds to_heatmap's result to decode (depth_train.py:395-399, 306-310,
500-504, 567-571; train.py:164-168), so to_heatmap here returns a light handle and decode runs one
kernel per (b, j) that does max, exp-sum and the three expectations in a single sweep.
"""
import numpy as np

from . import ops


class JointInfo:
    """utils.py:146-151"""

    def __init__(self, short_names, parent, mirror, key_index):
        self.short_names = short_names
        self.parent = parent
        self.mirror = mirror
        self.key_index = key_index


class VolumetricHeatmap:
    """Deferred softmax volume: what utils.to_heatmap returns here.  Holds the regressor output."""

    def __init__(self, logits, depth, num_joints, height, width):
        self.logits = logits
        self.depth, self.num_joints, self.height, self.width = depth, num_joints, height, width

    @property
    def shape(self):
        return (self.logits.shape[0], self.num_joints, self.height, self.width, self.depth)

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]


def to_heatmap(ausgabe, depth, num_joints, height, width):
    """ausgabe [B, depth*num_joints, H, W] (channel = d*J + j) -> handle of the [B,J,H,W,D] softmax volume."""
    if tuple(ausgabe.shape[1:]) != (depth * num_joints, height, width):
        raise ops.P3DError('to_heatmap: got %s, expected [B, %d, %d, %d]' % (tuple(ausgabe.shape), depth * num_joints, height, width))
    return VolumetricHeatmap(ausgabe, depth, num_joints, height, width)


def decode(heatmap, depth_range):
    """Soft-argmax expectation along x, y, z against linspace(0, 2, n), times depth_range -> [B, J, 3]."""
    if not isinstance(heatmap, VolumetricHeatmap):
        raise ops.P3DError('decode expects the handle returned by to_heatmap')
    return ops.softargmax3d(heatmap.logits, heatmap.depth, heatmap.num_joints, heatmap.height, heatmap.width, depth_range)


def get_info():
    """depth_main.get_info (depth_main.py:14-33): the H36M 17-joint JointInfo with index arrays."""
    from .joint_settings import h36m_base_joint, h36m_mirror, h36m_parent, h36m_short_names as names
    index = {name: i for i, name in enumerate(names)}
    mirror = np.array([index[h36m_mirror.get(n, n)] for n in names])
    parent = np.array([index[h36m_parent.get(n, n)] for n in names])
    return JointInfo(names, parent, mirror, index[h36m_base_joint])

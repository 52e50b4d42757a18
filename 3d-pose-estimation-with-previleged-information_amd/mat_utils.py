"""Image-space head of the legacy joint-space trainer (reference mat_utils.py:32-56, 84-118) on the fused HIP head kernel.

to_heatmap / decode: softmax over the H x W plane of each joint and its expectation against linspace(0, 1, n) * map_range.  That is the 3-D
head kernel (utils.decode) at depth 1: linspace(0, 2, n) * (map_range / 2) equals linspace(0, 1, n) * map_range exactly in binary floating point.
analyze / parse_epoch: the OKS-style image-space metrics, host numpy as in the reference.
"""
import numpy as np

from . import ops
from .utils import VolumetricHeatmap


def to_heatmap(ausgabe, num_joints, height, width):
    if tuple(ausgabe.shape[1:]) != (num_joints, height, width):
        raise ops.P3DError('mat_utils.to_heatmap: got %s, expected [B, %d, %d, %d]' % (tuple(ausgabe.shape), num_joints, height, width))
    return VolumetricHeatmap(ausgabe, 1, num_joints, height, width)


def decode(heatmap, map_range):
    if not isinstance(heatmap, VolumetricHeatmap) or heatmap.depth != 1:
        raise ops.P3DError('mat_utils.decode expects the handle returned by mat_utils.to_heatmap')
    coords = ops.softargmax3d(heatmap.logits, 1, heatmap.num_joints, heatmap.height, heatmap.width, map_range / 2.0)
    return coords[:, :, :2]


def coord_to_scale(true_mat, valid):
    """Longer side of the box around the valid joints of each pose (mat_utils.py:59-81)."""
    scales = []
    for points, mask in zip(true_mat, valid):
        points = points[mask]
        scales.append(max(points[:, 0].max() - points[:, 0].min(), points[:, 1].max() - points[:, 1].min()))
    return np.array(scales)


def analyze(spec_mat, true_mat, valid_mask, side_in):
    dist = np.linalg.norm(spec_mat - true_mat, axis=-1)
    scales = coord_to_scale(true_mat, valid_mask)
    oks = np.exp(-dist / np.expand_dims(2 * (scales / side_in) ** 2, axis=-1))
    oks = np.sum(oks * valid_mask, axis=-1) / np.sum(valid_mask, axis=-1)
    return dict(mat_mean=np.mean(dist[valid_mask]), score_oks=np.mean(oks), batch_size=spec_mat.shape[0])


def parse_epoch(scores):
    keys = ('score_oks', 'mat_mean')
    weights = np.array([patch['batch_size'] for patch in scores], dtype=np.float64)
    return {key: float(np.sum(weights * np.array([patch[key] for patch in scores])) / np.sum(weights)) for key in keys}

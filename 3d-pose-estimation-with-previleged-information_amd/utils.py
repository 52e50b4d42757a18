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


def get_deter_cam(spec_mat, relat_cam, intrinsics, valid=None):
    """Test-time numpy twin of get_recon_cam (utils.py:298-332): the same least-squares placement, on the host after evaluation."""
    spec_mat, relat_cam, intrinsics = (np.asarray(a) for a in (spec_mat, relat_cam, intrinsics))
    batch, joints = spec_mat.shape[:2]
    if valid is not None:
        count = np.sum(valid, axis=1)
        assert (count != 0).all() and (count != 1).all()
    unproject = np.linalg.inv(intrinsics).transpose(0, 2, 1)
    normalized = np.einsum('bij,bjk->bik', np.concatenate([spec_mat, np.ones((batch, joints, 1))], axis=-1), unproject)[:, :, :2]
    A = np.concatenate([np.tile(np.eye(2), (batch, joints, 1)), -normalized.reshape(batch, -1, 1)], axis=-1)
    b = (normalized * relat_cam[:, :, 2:] - relat_cam[:, :, :2]).reshape(batch, -1, 1)
    At = A.transpose(0, 2, 1)
    refer = np.linalg.inv(At @ A) @ (At @ b)
    return relat_cam + refer.transpose(0, 2, 1)


def get_recon_cam(spec_mat, relat_cam, intrinsics, valid=None):
    """Differentiable reconstruction of the reference-point location at train time (utils.py:335-366): least squares over ALL joints of
    relat_cam + t projecting onto spec_mat under `intrinsics`; returns relat_cam + t, [B, J, 3].

    The reference body asserts on a name `valid` that it never receives (utils.py:349-350, a NameError as shipped); here it is an optional
    argument with the same two checks (every sample needs at least two valid joints).  Like the reference, the fit itself does not use it."""
    if valid is not None:
        count = valid.sum(dim=1)
        assert bool((count != 0).all()) and bool((count != 1).all())
    return ops.recon_cam(spec_mat, relat_cam, intrinsics)


def get_attention(side_in, stride, image_coords, attention):
    """Distillation attention map (utils.py:14-42): sum over joints of exp(-r^2/5) around each joint's position on the
    side_out x side_out feature grid, normalised to max 1; all ones without -attention.  Host numpy, made in the loader."""
    side_out = (side_in - 1) // stride + 1
    if not attention:
        return np.ones((1, side_out, side_out))
    gx, gy = np.meshgrid(np.arange(side_out), np.arange(side_out))
    scale = side_in / side_out
    dist = (gx[..., None] - image_coords[:, 0] / scale) ** 2 + (gy[..., None] - image_coords[:, 1] / scale) ** 2
    radial = np.exp(-dist / 5.0).sum(axis=-1)
    return (radial / np.amax(radial))[None, :, :]


def get_info():
    """depth_main.get_info (depth_main.py:14-33): the H36M 17-joint JointInfo with index arrays."""
    from .joint_settings import h36m_base_joint, h36m_mirror, h36m_parent, h36m_short_names as names
    index = {name: i for i, name in enumerate(names)}
    mirror = np.array([index[h36m_mirror.get(n, n)] for n in names])
    parent = np.array([index[h36m_parent.get(n, n)] for n in names])
    return JointInfo(names, parent, mirror, index[h36m_base_joint])


# ---- evaluation metrics (reference utils.py:197-276); host-side numpy after the timed path, as in the reference ----------

def statistics(basic, flip, tangent, thresh):
    """Cascade of error classes over the valid joints (utils.py:197-224): each sample is counted by the FIRST test it passes
    -- solid (error <= thresh.solid), close (<= close), depth (image-plane error <= close), jitter (<= rough), switch (error against
    the mirrored joint <= rough) -- and what is left is `fail`.  Fractions of the total."""
    basic, flip, tangent = (np.asarray(a) for a in (basic, flip, tangent))
    count = float(basic.size)
    alive = np.ones(basic.shape, dtype=bool)
    out = {}
    for key, values, limit in (('solid', basic, thresh['solid']), ('close', basic, thresh['close']), ('depth', tangent, thresh['close']),
                               ('jitter', basic, thresh['rough']), ('switch', flip, thresh['rough'])):
        hit = alive & (values <= limit)
        out[key] = np.count_nonzero(hit) / count
        alive &= ~hit
    out['fail'] = np.count_nonzero(alive) / count
    return out


def analyze(spec_cam, true_cam, valid_mask, mirror, thresh):
    """Per-batch metrics (utils.py:237-276): mean joint error, PCK and AUC at thresh.rough, and the error-class fractions."""
    valid = np.asarray(valid_mask).reshape(-1).astype(bool)
    dist = np.linalg.norm(spec_cam - true_cam, axis=-1).reshape(-1)[valid]
    dist_flip = np.linalg.norm(spec_cam - true_cam[:, mirror], axis=-1).reshape(-1)[valid]
    dist_tangent = np.linalg.norm(spec_cam[:, :, :2] - true_cam[:, :, :2], axis=-1).reshape(-1)[valid]
    stats = statistics(dist, dist_flip, dist_tangent, thresh)
    stats.update(batch_size=dist.shape[0], score_pck=np.mean(dist / thresh['rough'] <= 1.0),
                 score_auc=np.mean(np.maximum(0, 1 - dist / thresh['rough'])), cam_mean=np.mean(dist))
    return stats


def parse_epoch(stats):
    """Valid-joint-count weighted average of the per-batch dicts (utils.py:227-234)."""
    keys = ('solid', 'close', 'jitter', 'depth', 'switch', 'fail', 'score_pck', 'score_auc', 'cam_mean')
    weights = np.array([patch['batch_size'] for patch in stats], dtype=np.float64)
    return {key: float(np.sum(weights * np.array([patch[key] for patch in stats])) / np.sum(weights)) for key in keys}

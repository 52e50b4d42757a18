"""Deterministic synthetic weights and batches (numpy only, no torch, no GPU).

There is no dataset and no checkpoint on the build or GPU box, and the reference
ships no fixtures.  Everything that needs numbers --
the golden-vector generator that drives the real reference, the oracle, the
parity tests, smoke() and bench.py -- draws them from this one generator so the
same tensors can be re-created anywhere from a seed instead of being committed.

Weight rule (follows the reference initialisers, with non-trivial BN affine
parameters so that gamma/beta paths are exercised):
  * conv weight without bias: N(0, 2/(k*k*Cout))  -- kaiming_normal_(fan_out, relu),
    reference depthnet.py:148-150, resnet.py:151-154, fusionnet.py:186-189
  * conv with bias (the regressors, created after the init loop, depthnet.py:156):
    torch default U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias
  * BatchNorm weight: 1 + 0.1*N(0,1); bias: 0.1*N(0,1) (reference uses 1 / 0; the
    perturbation keeps the affine terms from being invisible to parity tests)
  * running_mean 0, running_var 1, num_batches_tracked 0 (torch defaults)

Batch rule: SURVEY.md section 8d (color ~ N(0,1); depth ~ U[0,1) with <0.3 -> 0;
true_cam ~ N(0, 300^2) mm; true_val all True or ~20 % False), seed = 1000*rank + step.
"""
import zlib

import numpy as np

REGRESSOR_KEYS = ("regressor", "cam_regressor", "mat_regressor")


def _rng(name, seed):
    return np.random.Generator(np.random.PCG64([zlib.crc32(name.encode()), int(seed)]))


def _is_regressor(name):
    head = name.split(".")[0]
    return head in REGRESSOR_KEYS


def det_tensor(name, shape, seed=0):
    """One deterministic float32 (or int64 for num_batches_tracked) array for a state-dict key."""
    shape = tuple(int(s) for s in shape)
    rng = _rng(name, seed)
    leaf = name.split(".")[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "running_mean":
        return np.zeros(shape, dtype=np.float32)
    if leaf == "running_var":
        return np.ones(shape, dtype=np.float32)
    if len(shape) == 4:
        cout, cin, kh, kw = shape
        if _is_regressor(name):
            bound = 1.0 / np.sqrt(cin * kh * kw)
            return rng.uniform(-bound, bound, size=shape).astype(np.float32)
        std = np.sqrt(2.0 / (kh * kw * cout))
        return (rng.standard_normal(size=shape) * std).astype(np.float32)
    if len(shape) == 1:
        if _is_regressor(name):
            # bias of a regressor conv; fan_in is not recoverable from the bias shape, use a fixed small bound
            return rng.uniform(-0.01, 0.01, size=shape).astype(np.float32)
        if leaf == "weight":
            return (1.0 + 0.1 * rng.standard_normal(size=shape)).astype(np.float32)
        if leaf == "bias":
            return (0.1 * rng.standard_normal(size=shape)).astype(np.float32)
    raise ValueError("no deterministic rule for %s %r" % (name, shape))


def det_state_dict(shapes, seed=0):
    """shapes: ordered mapping key -> shape (e.g. from model.state_dict()).  Returns key -> ndarray."""
    return {k: det_tensor(k, s, seed) for k, s in shapes.items()}


def make_batch(batch, side=256, num_joints=17, rank=0, step=0, invalid_frac=0.0, depth_holes=0.3):
    """Synthetic train tuple with the depth_datasets.py:236-237 contract.

    Returns (color[B,3,S,S] f32, depth[B,1,S,S] f32, true_cam[B,J,3] f32, true_val[B,J] bool).
    """
    rng = np.random.Generator(np.random.PCG64(1000 * int(rank) + int(step)))
    color = rng.standard_normal(size=(batch, 3, side, side), dtype=np.float32)
    depth = rng.random(size=(batch, 1, side, side), dtype=np.float32)
    depth[depth < depth_holes] = 0.0
    true_cam = (rng.standard_normal(size=(batch, num_joints, 3)) * 300.0).astype(np.float32)
    if invalid_frac > 0.0:
        true_val = rng.random(size=(batch, num_joints)) >= invalid_frac
        true_val[:, -1] = True
    else:
        true_val = np.ones((batch, num_joints), dtype=bool)
    return color, depth, true_cam, true_val

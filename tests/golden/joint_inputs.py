"""Seeded inputs of the legacy joint-space iteration, shared by make_golden.py (reference side) and tests/test_joint_space.py."""
import numpy as np


def joint_batch(synth, batch, side, seed):
    """Inputs of one legacy joint-space iteration (train.py:66): image, true_cam, true_mat, true_val, intrinsics."""
    c, d, tc, tv = synth.make_batch(batch, side=side, rank=5, step=seed)
    rng = np.random.Generator(np.random.PCG64(900 + seed))
    intr = np.tile(np.array([[1.2 * side, 0, side / 2], [0, 1.2 * side, side / 2], [0, 0, 1]], np.float32), (batch, 1, 1))
    intr[:, 0, 0] *= rng.uniform(0.9, 1.1, batch).astype(np.float32)
    intr[:, 0, 2] += rng.uniform(-5, 5, batch).astype(np.float32)
    cam = tc.copy()
    cam[:, :, 2] += 3000.0                                                      # in front of the camera
    proj = cam[:, :, :2] / cam[:, :, 2:] * intr[:, None, [0, 1], [0, 1]] + intr[:, None, :2, 2]
    true_mat = (proj + rng.standard_normal(proj.shape) * 2).astype(np.float32)
    tv = tv.copy()
    tv[:, :2] = True
    return c, cam.astype(np.float32), true_mat, tv, intr

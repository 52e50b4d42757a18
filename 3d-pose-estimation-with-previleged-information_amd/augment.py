"""On-GPU batch versions of the reference's per-image CPU augmentations (BASELINE config 5):
augment_colour.random_color (augment_colour.py:48-67) and augment_occluder.random_erase (augment_occluder.py:84-105).

The reference runs them in DataLoader worker processes on HWC uint8 images with cv2; here a whole batch
[B,3,H,W] of 0..255 float values is transformed in place by one kernel launch each, the random draws being made
on the host with the reference's distributions.  (random_occlu needs an occluder image bank on disk and cv2.resize
and is dead code in the reference -- no caller -- so it is not provided.)
"""
import numpy as np
import torch

from . import ops


def draw_colour_params(batch, rng):
    """brightness U(-0.125, 0.125), contrast U(0.8, 1.25), hue U(-18, 18) degrees, saturation U(0.8, 1.25)."""
    return np.stack([rng.uniform(-0.125, 0.125, batch), rng.uniform(0.8, 1.25, batch), rng.uniform(-18, 18, batch),
                     rng.uniform(0.8, 1.25, batch)], axis=1).astype(np.float32)


def draw_erase_rects(batch, height, width, rng):
    """Rectangle of area U(0.1, 0.25)*H*W, aspect U(0.4, 2.5), uniformly placed, random 0..255 colour (augment_occluder.py:84-101)."""
    rects = np.zeros((batch, 4), dtype=np.int32)
    for i in range(batch):
        area = rng.uniform(0.1, 0.25) * height * width
        aspect = rng.uniform(0.4, 2.5)
        eh, ew = (area * aspect) ** 0.5, (area / aspect) ** 0.5
        start = (np.array([height, width]) - np.array([eh, ew])) * rng.uniform(size=2)
        end = start + np.array([eh, ew])
        y0, x0 = np.round(start).astype(int)
        y1, x1 = np.round(end).astype(int)
        rects[i] = (x0, y0, x1, y1)
    colour = rng.integers(0, 256, size=(batch, 3)).astype(np.float32)
    return rects, colour


def random_color_(images255, rng):
    """In place on a contiguous [B,3,H,W] fp32 device tensor holding 0..255 values."""
    params = torch.from_numpy(draw_colour_params(images255.shape[0], rng)).to(images255.device)
    return ops.augment_colour_(images255, params)


def random_erase_(images255, rng):
    b, c, h, w = images255.shape
    rects, colour = draw_erase_rects(b, h, w, rng)
    return ops.augment_erase_(images255, torch.from_numpy(rects).to(images255.device), torch.from_numpy(colour).to(images255.device))


class GpuAugment:
    """What the reference's loader does to the colour image of a TRAINING sample after cropping (depth_datasets.py:210):
    `transform(random_color(image) if colour else image)`, here for a whole batch on the GPU: the loader hands over raw 0..255
    crops, this applies colour jitter (-colour), random erasing (-eraser) and ToTensor + Normalize, in place, three launches."""

    def __init__(self, colour, eraser, seed=0):
        self.colour, self.eraser = bool(colour), bool(eraser)
        self.rng = np.random.Generator(np.random.PCG64(seed))

    def __call__(self, images255, train=True):
        if train and self.colour:
            random_color_(images255, self.rng)
        if train and self.eraser:
            random_erase_(images255, self.rng)
        return ops.normalize_rgb_(images255)

"""On-GPU batch versions of the reference's per-image CPU augmentations (BASELINE config 5):
augment_colour.random_color (augment_colour.py:48-67) and augment_occluder.random_erase (augment_occluder.py:84-105).

The reference runs them in DataLoader worker processes on HWC uint8 images with cv2; here a whole batch
[B,3,H,W] of 0..255 float values is transformed in place by one kernel launch each, the random draws being made
on the host with the reference's distributions.  augment_occluder.paste_over / random_occlu (augment_occluder.py:7-81, no caller in the
reference) are here too: the clipped paste rectangles are planned on the host (`plan_paste`, the reference's own index arithmetic) and one
launch blends every image's occluder in (`paste_over_`, pinned to the reference's paste_over: tests/golden/augment.npz); the occluder bank of
`random_occlu_` is resized on the host by an area average restated from cv2.INTER_AREA (cv2 is absent here: that resize is parity-unpinned).
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


def plan_paste(occ_shape, image_shape, center):
    """The index arithmetic of augment_occluder.paste_over (augment_occluder.py:27-50) -> (dst_y0, dst_x0, src_y0, src_x0, h, w).
    The reference slices with float bounds, which the numpy it was written for truncated to integers (and numpy >= 1.12 rejects); the
    truncation is reproduced.  Where the two truncated extents differ (odd-sized occluders cut by the border: the reference's own
    assignment fails to broadcast there) the common part is pasted."""
    shape_occ = np.array(occ_shape[:2])
    shape_image = np.array(image_shape[:2])
    center = np.round(center).astype(int)
    ideal_start_dst = center - shape_occ / 2
    ideal_end_dst = ideal_start_dst + shape_occ
    start_dst = np.maximum(ideal_start_dst, 0)
    end_dst = np.minimum(ideal_end_dst, shape_image)
    start_src = start_dst - ideal_start_dst
    end_src = shape_occ + (end_dst - ideal_end_dst)
    d0, d1 = [int(v) for v in start_dst], [int(v) for v in end_dst]
    s0, s1 = [int(v) for v in start_src], [int(v) for v in end_src]
    h = max(min(d1[0] - d0[0], s1[0] - s0[0]), 0)
    w = max(min(d1[1] - d0[1], s1[1] - s0[1]), 0)
    return d0[0], d0[1], s0[0], s0[1], h, w


def paste_over_(images255, occluders, alphas, centers, truncate=True):
    """Batch augment_occluder.paste_over, in place on a contiguous [B,C,H,W] fp32 device tensor holding 0..255 values.  occluders: one
    [h,w,C] array per image (or None: leave that image alone), alphas: one [h,w] array in 0..1 per image or None (opaque), centers [B,2] (row, col)."""
    b, c, h, w = images255.shape
    plan = np.zeros((b, 8), dtype=np.int32)
    bank, alpha, offset, any_alpha = [], [], 0, any(a is not None for a in alphas)
    for i in range(b):
        occ = occluders[i]
        if occ is None:
            continue
        occ = np.asarray(occ, dtype=np.float32).reshape(occ.shape[0], occ.shape[1], -1)
        plan[i, :6] = plan_paste(occ.shape, (h, w), centers[i])
        plan[i, 6], plan[i, 7] = occ.shape[1], offset
        bank.append(occ.reshape(-1, occ.shape[2]))
        if any_alpha:
            alpha.append(np.ones(occ.shape[:2], np.float32).reshape(-1) if alphas[i] is None else np.asarray(alphas[i], np.float32).reshape(-1))
        offset += occ.shape[0] * occ.shape[1]
    if not bank:
        return images255
    dev = images255.device
    bank_t = torch.from_numpy(np.concatenate(bank)).to(dev)
    alpha_t = torch.from_numpy(np.concatenate(alpha)).to(dev) if any_alpha else None
    return ops.augment_occlude_(images255, bank_t, alpha_t, torch.from_numpy(plan).to(dev), int((plan[:, 4] * plan[:, 5]).max()), truncate)


def resize_area(image, dest_hw):
    """cv2.resize(image, dest[::-1], interpolation=cv2.INTER_AREA) for a down-scale, restated as the exact box filter: every destination
    pixel is the area-weighted mean of the source pixels its footprint covers.  cv2 is absent here: parity unpinned."""
    src = np.asarray(image, dtype=np.float64)
    squeeze = src.ndim == 2
    src = src.reshape(src.shape[0], src.shape[1], -1)

    def weights(n_src, n_dst):
        scale = n_src / n_dst
        m = np.zeros((n_dst, n_src))
        for o in range(n_dst):
            lo, hi = o * scale, (o + 1) * scale
            for i in range(int(np.floor(lo)), min(int(np.ceil(hi)), n_src)):
                m[o, i] = max(min(hi, i + 1) - max(lo, i), 0.0) / scale
        return m
    wy, wx = weights(src.shape[0], dest_hw[0]), weights(src.shape[1], dest_hw[1])
    out = np.einsum('oi,ijc,pj->opc', wy, src, wx)
    if np.issubdtype(np.asarray(image).dtype, np.integer):
        out = np.clip(np.rint(out), 0, 255)
    out = out.astype(np.asarray(image).dtype)
    return out[:, :, 0] if squeeze else out


def random_occlu_(images255, occluder_bank, rng):
    """Batch augment_occluder.random_occlu (augment_occluder.py:68-81): per image one occluder of `occluder_bank` (a list of (occluder [h,w,3],
    mask [h,w]) pairs, the contents of the reference's occluder_i.npy / mask_i.npy files) scaled by U(0.4, 0.8) and pasted at a uniform centre."""
    b, c, h, w = images255.shape
    occs, alphas, centers = [], [], np.zeros((b, 2))
    for i in range(b):
        occluder, occ_mask = occluder_bank[rng.integers(len(occluder_bank))]
        dest = tuple(np.round(rng.uniform(0.4, 0.8) * np.array(occluder.shape[:2])).astype(int))
        occs.append(resize_area(occluder, dest))
        alphas.append(resize_area(occ_mask, dest))
        centers[i] = np.array([h, w]) * rng.uniform(size=2)
    return paste_over_(images255, occs, alphas, centers)


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

    def __init__(self, colour, eraser, seed=0, occluder_bank=None):
        self.colour, self.eraser = bool(colour), bool(eraser)
        self.occluder_bank = occluder_bank          # list of (occluder, mask) arrays: BASELINE config 5's augment_occluder leg
        self.rng = np.random.Generator(np.random.PCG64(seed))

    def __call__(self, images255, train=True):
        if train and self.colour:
            random_color_(images255, self.rng)
        if train and self.occluder_bank:
            random_occlu_(images255, self.occluder_bank, self.rng)
        if train and self.eraser:
            random_erase_(images255, self.rng)
        return ops.normalize_rgb_(images255)

"""Person-centred crops for the file-backed loaders (reference depth_datasets.py:153-237, datasets.py:83-148), split for the GPU:

  worker processes (CPU)   decode the frame, plan the virtual camera (`plan_crop`: a handful of 3x3 products), transform the labels;
  main process (GPU)       `GpuCropLoader` uploads the raw frames of a batch and resamples all crops with one `p3d_reproject_crops` launch
                           per stream, then depth enhancement / ToTensor + Normalize kernels -- the work cv2.remap and torchvision do per
                           sample inside the reference's DataLoader workers.

The loader the trainers see keeps the reference's contract: len() and an iterator of (color, depth, true_cam, true_val[, back_rotate | atten_map])
tuples (depth_datasets.py:227-237) or (color, true_cam, true_val[, back_rotate]) (datasets.py:143-148), the image tensors already on the device.
"""
import numpy as np
import torch

from . import cameralib

MEAN = (0.485, 0.456, 0.406)          # depth_datasets.py:77-78
DEV = (0.229, 0.224, 0.225)


def imread(path):
    """matplotlib.pyplot.imread's convention (what depth_datasets.py:193 relies on): PNG -> fp32 in [0, 1] (8-bit / 255, 16-bit / 65535),
    anything else -> the decoder's uint8 array."""
    from PIL import Image
    with Image.open(path) as image:
        is_png = (image.format or '').upper() == 'PNG'
        if image.mode == 'P':
            image = image.convert('RGBA' if 'transparency' in image.info else 'RGB')
        data = np.asarray(image)
    if not is_png:
        return data
    if data.dtype == np.uint8:
        return data.astype(np.float32) / 255
    if data.dtype == bool:
        return data.astype(np.float32)
    return np.divide(data.astype(np.float32), 2 ** 16 - 1, dtype=np.float32)          # 'I;16' / 'I' depth frames


def plan_crop(camera, bbox, side_in, zoom=None, do_flip=False):
    """Virtual camera of get_input_image (depth_datasets.py:162-191): look at the box centre, drop the lens distortion, square the pixels, zoom so
    that the LONGER box side spans side_in pixels, principal point at the crop centre, optional extra zoom (-geometry) and mirror."""
    bbox = np.asarray(bbox, np.float64)
    center = bbox[:2] + bbox[2:] / 2
    half = np.array([bbox[2] / 2, 0]) if bbox[2] >= bbox[3] else np.array([0, bbox[3] / 2])
    far_side = np.stack([center - half, center + half])
    new_cam = camera.copy()
    new_cam.turn_towards(center)
    new_cam.undistort()
    new_cam.square_pixels()
    far_side = new_cam.world_to_image(camera.image_to_world(far_side))
    new_cam.zoom(side_in / np.linalg.norm(far_side[0] - far_side[1]))
    new_cam.center_principal_point((side_in, side_in))
    if zoom is not None:
        new_cam.zoom(zoom)
    if do_flip:
        new_cam.horizontal_flip()
    return new_cam


def frame_and_params(path, camera, new_cam):
    """(frame [H,W,C] uint8 | fp32, params20, round flag): uint8 sources are resampled to rounded uint8 values like cv2.remap; 8-bit PNGs, which
    the reference resamples as fp32 in [0, 1], travel as their uint8 codes and are resampled unrounded (the / 255 happens with ToTensor)."""
    image = imread(path)
    round_u8 = image.dtype == np.uint8
    if image.dtype != np.uint8 and image.ndim == 3:                               # 8-bit colour PNG: exact uint8 codes, 4x fewer bytes to upload
        image = np.rint(image[:, :, :3] * 255).astype(np.uint8)
    if image.ndim == 3 and image.shape[2] == 4:
        image = image[:, :, :3]
    frame = np.array(image.reshape(image.shape[0], image.shape[1], -1))              # own, writable copy (PIL hands out read-only views)
    return frame, cameralib.reproject_params(camera, new_cam), round_u8


def to_depth_divisor(depth_cam, side_in):
    """utils.to_depth's divisor over a side_in x side_in crop (utils.py:68-75): sqrt(|image_to_camera(u, v)|^2 + 1), with the ORIGINAL depth camera."""
    u, v = np.meshgrid(range(side_in), range(side_in))
    rays = depth_cam.image_to_camera(np.stack([u, v], axis=-1).reshape(-1, 2)).reshape(side_in, side_in, -1)
    return np.sqrt(np.sum(rays.astype(np.float64) ** 2, axis=-1) + 1).astype(np.float32)


class GpuCropLoader:
    """Wraps the DataLoader over raw samples (dicts, see depth_datasets.Dataset.parse_sample) and yields the reference's tuples.

    The GPU stage of batch i+1 (pinned upload of the raw frames, ~8 ms for 64 1080p frames, + the crop kernels) is issued on a second HIP stream while
    the trainer works on batch i on the launch stream, so the copy engine runs beside the convolutions; the consumer side only waits on an event."""

    def __init__(self, loader, side_in, raw_color, device=None, prefetch=True):
        self.loader, self.side_in, self.raw_color = loader, side_in, raw_color
        self.device = device
        self.dataset = loader.dataset
        self.prefetch = prefetch
        self._stream = None

    def __len__(self):
        return len(self.loader)

    def _crops(self, frames, params, rounded):
        from . import ops
        frames = frames.to(self.device, non_blocking=True)
        params = params.to(self.device, non_blocking=True)
        flags = set(bool(flag) for flag in rounded)
        if len(flags) != 1:
            raise ValueError('a batch mixes uint8 and fp32 source frames')
        return ops.reproject_crops(frames, params, (self.side_in, self.side_in), round_u8=flags.pop())

    def _stage(self, raw):
        """Everything the GPU does for one batch, on the CURRENT stream -> the reference's tuple."""
        from . import ops
        color = self._crops(raw['color_frame'], raw['color_params'], raw['color_round'])
        if not self.raw_color:
            ops.normalize_rgb_(color)                                          # ToTensor (/255) + Normalize (depth_datasets.py:90-92)
        items = [color]
        if 'depth_frame' in raw:
            depth = self._crops(raw['depth_frame'], raw['depth_params'], raw['depth_round'])
            divisor = raw['depth_divisor'].to(self.device).reshape(depth.shape).contiguous() if 'depth_divisor' in raw else None
            ops.enhance_depth_(depth, float(raw['depth_threshold'][0]), bool(raw['nexponent'][0]), divisor)
            items.append(depth)
        items.append(raw['true_cam'])
        if 'true_mat' in raw:                                                      # joint-space tuples of the legacy trainer (train.py:66)
            items.append(raw['true_mat'])
        items.append(raw['true_val'])
        for key in ('intrinsics', 'back_rotate', 'atten_map'):
            if key in raw:
                items.append(raw[key])
        return tuple(items)

    def _stage_ahead(self, raw):
        with torch.cuda.stream(self._stream):
            items = self._stage(raw)
            ready = torch.cuda.Event()
            ready.record(self._stream)
        return items, ready

    def __iter__(self):
        if self.device is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        if not self.prefetch:
            for raw in self.loader:
                yield self._stage(raw)
            return
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.device)
        source = iter(self.loader)
        ahead = None
        for raw in source:
            staged = self._stage_ahead(raw)
            if ahead is not None:
                yield self._hand_over(ahead)
            ahead = staged
        if ahead is not None:
            yield self._hand_over(ahead)

    def _hand_over(self, staged):
        items, ready = staged
        consumer = torch.cuda.current_stream(self.device)
        consumer.wait_event(ready)
        for item in items:
            if torch.is_tensor(item) and item.is_cuda:
                item.record_stream(consumer)                                       # allocated on the side stream, used on the launch stream
        return items

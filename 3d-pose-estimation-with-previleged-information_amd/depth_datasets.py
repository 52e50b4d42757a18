"""Loader API of the reference (depth_datasets.py:23-28):
    data_loader(args, phase, data_info) -> loader with len() and an iterator
yielding, for training, (color[B,3,S,S] f32, depth[B,1,S,S] f32, true_cam[B,J,3] f32, true_val[B,J] bool) (depth_datasets.py:236-237), for
validation additionally back_rotate[B,3,3] (:227-229), under -do_teach additionally atten_map (:231-234).

File-backed datasets (`-data_name ntu | pku`, site layout of depth_datasets.py:95-150 under metadata.json's root): the worker processes read the
sample / camera pickles written by the reference's preprocessing, decode the colour and depth frames and plan the crop cameras; the crops themselves
are resampled on the GPU for the whole batch (crops.GpuCropLoader -> p3d_reproject_crops, p3d_enhance_depth, p3d_normalize_rgb), so the image tensors
arrive already on the device.  With `-synthetic N` the loader serves N deterministic synthetic batches per epoch instead (synth.make_batch rules),
which is what the step-parity tests, smoke() and bench.py use.
"""
import glob
import json
import os

import numpy as np
import torch
import torch.utils.data as data

from . import cameralib, crops, synth


def data_loader(args, phase, data_info):
    dataset = Dataset(args, phase, data_info)
    loader = data.DataLoader(dataset, args.batch_size, shuffle=(args.shuffle and phase == 'train'), num_workers=args.workers, pin_memory=True)
    if dataset.synthetic:
        return loader
    return crops.GpuCropLoader(loader, args.side_in, dataset.raw_color)


def ntu_split(split, phase, sample):
    return (sample['video'][:8] in split[phase]['configs']) and (sample['video'][8:12] in split[phase]['persons'])      # depth_datasets.py:31-32


def pku_split(split, phase, sample):
    return sample['video'] in split[phase]


ENHANCE_THRESHOLD = dict(ntu=0.1, pku=0.5)                # enhance_ntu / enhance_pku (depth_datasets.py:39-56)


def shard(samples, phase):
    """One process per GPU: under torchrun each rank trains on every WORLD_SIZE-th sample (nn.DataParallel split each batch instead)."""
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if world > 1 and phase == 'train':
        # equal shard lengths: a rank with one batch more than the others would wait forever in the gradient all-reduce
        return samples[:len(samples) // world * world][rank::world]
    return samples


class Dataset(data.Dataset):

    def __init__(self, args, phase, data_info):
        assert phase in ('train', 'valid', 'test')
        assert len(data_info.short_names) == args.num_joints                      # depth_datasets.py:62
        self.phase = phase
        self.data_info = data_info
        self.side_in = args.side_in
        self.num_joints = args.num_joints
        self.at_test = phase != 'train'
        self.do_teach = bool(getattr(args, 'do_teach', False)) and phase == 'train'
        self.stride, self.attention = args.stride, args.attention
        # -colour / -eraser: the colour stream is handed over RAW (0..255 values); augmentation + ToTensor/Normalize run on the GPU
        # in the trainer (augment.GpuAugment) instead of in the loader workers (depth_datasets.py:210)
        self.raw_color = bool(args.colour or args.eraser)
        self.synthetic = int(getattr(args, 'synthetic', 0) or 0)
        if self.synthetic:
            self.count = self.synthetic * args.batch_size
            return
        from .depth_train import _load_metadata
        metadata = _load_metadata(args)
        if not metadata or args.data_name not in metadata.get('root', {}):
            raise FileNotFoundError('no dataset root for %r: give -metadata <metadata.json with root[%s]> or -synthetic N' % (args.data_name, args.data_name))
        if args.data_name not in ENHANCE_THRESHOLD:
            raise ValueError('depth_datasets serves the ntu and pku layouts (depth_datasets.py:95-150); got -data_name %s' % args.data_name)
        self.data_name = args.data_name
        self.root = metadata['root'][args.data_name]
        self.samples = shard(getattr(self, 'get_' + args.data_name + '_samples')(phase, globals()[args.data_name + '_split']), phase)
        getattr(self, 'init_' + args.data_name)()
        self.nexponent = args.nexponent
        self.geometry = args.geometry and (not self.at_test)
        self.random_zoom = args.random_zoom
        self.to_depth = args.to_depth
        self._divisors = {}

    # ---- site layout (depth_datasets.py:95-150) ----
    def init_ntu(self):
        self.depth_cams = cameralib.load_pickle(os.path.join(self.root, 'depth_cameras.pkl'))

    def init_pku(self):
        self.cameras = cameralib.load_pickle(os.path.join(self.root, 'cameras.pkl'))

    def depth_cam_ntu(self, sample):
        return self.depth_cams[sample['video'][:8]]

    def depth_cam_pku(self, sample):
        return self.cameras[sample['video'][5]]

    def depth_image_ntu(self, sample):
        folder = os.path.join('nturgbd_depth_s' + sample['video'][1:4], 'nturgb+d_depth')
        return os.path.join(self.root, folder, sample['video'], 'Depth-' + str(sample['frame'] + 1).zfill(8) + '.png')

    def depth_image_pku(self, sample):
        return os.path.join(self.root, 'DEPTH_IMAGE', sample['video'] + '.' + str(sample['frame']) + '.png')

    def _split(self):
        with open(os.path.join(self.root, 'split.json')) as file:
            return json.load(file)

    def get_ntu_samples(self, phase, split_by):
        samples = []
        for sample_file in sorted(glob.glob(os.path.join(self.root, 'final_samples', '*.pkl'))):
            samples += cameralib.load_pickle(sample_file)
        split = self._split()
        return [sample for sample in samples if split_by(split, phase, sample)]

    def get_pku_samples(self, phase, split_by):
        samples = cameralib.load_pickle(os.path.join(self.root, 'final_samples.pkl'))
        split = self._split()
        return [sample for sample in samples if split_by(split, phase, sample)]

    # ---- one sample (depth_datasets.py:153-237) ----
    def get_input_image(self, image_path, camera, bbox, do_flip, random_zoom):
        """-> (frame, new_cam, params20, round flag): the decoded frame and the crop camera; the resampling happens on the GPU per batch."""
        new_cam = crops.plan_crop(camera, bbox, self.side_in, random_zoom if self.geometry else None, do_flip)
        frame, params, rounded = crops.frame_and_params(image_path, camera, new_cam)
        return frame, new_cam, params, rounded

    def parse_sample(self, sample):
        depth_cam = getattr(self, 'depth_cam_' + self.data_name)(sample)
        depth_path = getattr(self, 'depth_image_' + self.data_name)(sample)
        do_flip = (not self.at_test) and (np.random.rand() < 0.5)
        random_zoom = np.random.uniform(self.random_zoom, self.random_zoom ** (-1))
        color_frame, new_color_cam, color_params, color_round = self.get_input_image(sample['image'], sample['camera'], sample['bbox'], do_flip, random_zoom)
        depth_frame, _, depth_params, depth_round = self.get_input_image(depth_path, depth_cam, sample['depth_bbox'], do_flip, random_zoom)
        camera_coords = new_color_cam.world_to_camera(sample['skeleton'])
        valid = np.asarray(sample['valid'])
        if do_flip:
            camera_coords = camera_coords[self.data_info.mirror]
            valid = valid[self.data_info.mirror]
        out = dict(color_frame=torch.from_numpy(color_frame), color_params=torch.from_numpy(color_params), color_round=color_round,
                   depth_frame=torch.from_numpy(depth_frame), depth_params=torch.from_numpy(depth_params), depth_round=depth_round,
                   depth_threshold=ENHANCE_THRESHOLD[self.data_name], nexponent=bool(self.nexponent),
                   true_cam=torch.from_numpy(np.ascontiguousarray(camera_coords, dtype=np.float32)),
                   true_val=torch.from_numpy(np.ascontiguousarray(valid, dtype=bool)))
        if self.to_depth:
            key = id(depth_cam)
            if key not in self._divisors:
                self._divisors[key] = torch.from_numpy(crops.to_depth_divisor(depth_cam, self.side_in))
            out['depth_divisor'] = self._divisors[key]
        if self.at_test:
            out['back_rotate'] = torch.from_numpy(np.asarray(sample['camera'].R @ new_color_cam.R.T, dtype=np.float32))
        elif self.do_teach:
            from .utils import get_attention
            image_coords = new_color_cam.camera_to_image(camera_coords)
            out['atten_map'] = torch.from_numpy(np.asarray(get_attention(self.side_in, self.stride, image_coords, self.attention))).float()
        return out

    def __len__(self):
        return self.count if self.synthetic else len(self.samples)

    def __getitem__(self, index):
        if not self.synthetic:
            return self.parse_sample(self.samples[index])
        color, depth, cam, val = synth.make_batch(1, side=self.side_in, num_joints=self.num_joints, rank=0, step=index)
        if self.raw_color:
            raw = np.random.Generator(np.random.PCG64(7919 + index)).integers(0, 256, size=color[0].shape)
            color = raw[None].astype(np.float32)
        items = [torch.from_numpy(color[0]), torch.from_numpy(depth[0]), torch.from_numpy(cam[0]), torch.from_numpy(val[0])]
        if self.at_test:
            items.append(torch.eye(3))
        if self.do_teach:                      # (color, depth, cam, valid, atten_map): depth_datasets.py:231-234
            from .utils import get_attention
            coords = np.random.Generator(np.random.PCG64(index)).uniform(0, self.side_in, size=(self.num_joints, 2))
            items.append(torch.from_numpy(get_attention(self.side_in, self.stride, coords, self.attention)).float())
        return tuple(items)

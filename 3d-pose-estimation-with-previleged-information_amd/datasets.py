"""RGB-only loader API of the legacy entry point (datasets.py:22-27,141-148):
    data_loader(args, phase, data_info) -> loader with len() and an iterator
yielding (color[B,3,S,S] f32, cam[B,J,3] f32, valid[B,J] bool[, back_rotate[B,3,3]]).  Under -joint_space the tuples carry what train.py's joint_train /
joint_test unpack (train.py:66,212) and the reference's own datasets.py no longer produces: true_mat[B,J,2] = the joints projected into the crop and
intrinsics[B,3,3] of the crop camera -> (color, cam, mat, valid, intrinsics[, back_rotate]).

File-backed mode (`-data_name h36m`: <root>/samples.pkl + split.json, datasets.py:30-72): as in depth_datasets, the workers decode frames and plan
the crop cameras and the crops are resampled on the GPU per batch (crops.GpuCropLoader).  `-synthetic N` serves N deterministic synthetic batches.
"""
import json
import os

import numpy as np
import torch
import torch.utils.data as data

from . import cameralib, crops, synth
from .depth_datasets import shard


def data_loader(args, phase, data_info):
    dataset = Dataset(args, phase, data_info)
    loader = data.DataLoader(dataset, args.batch_size, shuffle=(args.shuffle and phase == 'train'), num_workers=args.workers, pin_memory=True)
    if dataset.synthetic:
        return loader
    return crops.GpuCropLoader(loader, args.side_in, dataset.raw_color)


def h36m_split(split, phase, sample):
    folder = os.path.basename(os.path.dirname(sample['image']))                   # datasets.py:30-33
    return folder.split('.')[0] in split[phase]


class Dataset(data.Dataset):

    def __init__(self, args, phase, data_info):
        assert phase in ('train', 'valid', 'test')
        assert len(data_info.short_names) == args.num_joints
        self.phase = phase
        self.data_info = data_info
        self.side_in = args.side_in
        self.num_joints = args.num_joints
        self.at_test = phase != 'train'
        self.raw_color = bool(args.colour or args.eraser)
        self.joint_space = bool(getattr(args, 'joint_space', False))
        self.synthetic = int(getattr(args, 'synthetic', 0) or 0)
        if self.synthetic:
            self.count = self.synthetic * args.batch_size
            return
        from .depth_train import _load_metadata
        metadata = _load_metadata(args)
        if not metadata or args.data_name not in metadata.get('root', {}):
            raise FileNotFoundError('no dataset root for %r: give -metadata <metadata.json with root[%s]> or -synthetic N' % (args.data_name, args.data_name))
        if args.data_name != 'h36m':
            raise ValueError('datasets serves the h36m layout (datasets.py:64-72); got -data_name %s' % args.data_name)
        self.data_name = args.data_name
        self.root = metadata['root'][args.data_name]
        self.samples = shard(self.get_h36m_samples(phase, h36m_split), phase)
        self.geometry = args.geometry and (not self.at_test)
        self.random_zoom = args.random_zoom

    def get_h36m_samples(self, phase, split_by):
        samples = cameralib.load_pickle(os.path.join(self.root, 'samples.pkl'))
        with open(os.path.join(self.root, 'split.json')) as file:
            split = json.load(file)
        return [sample for sample in samples if split_by(split, phase, sample)]

    def parse_sample(self, sample):
        do_flip = (not self.at_test) and (np.random.rand() < 0.5)                # datasets.py:124-148
        random_zoom = np.random.uniform(self.random_zoom, self.random_zoom ** (-1))
        new_cam = crops.plan_crop(sample['camera'], sample['bbox'], self.side_in, random_zoom if self.geometry else None, do_flip)
        frame, params, rounded = crops.frame_and_params(sample['image'], sample['camera'], new_cam)
        camera_coords = new_cam.world_to_camera(sample['skeleton'])
        valid = np.asarray(sample['valid'])
        if do_flip:
            camera_coords = camera_coords[self.data_info.mirror]
            valid = valid[self.data_info.mirror]
        out = dict(color_frame=torch.from_numpy(frame), color_params=torch.from_numpy(params), color_round=rounded,
                   true_cam=torch.from_numpy(np.ascontiguousarray(camera_coords, dtype=np.float32)),
                   true_val=torch.from_numpy(np.ascontiguousarray(valid, dtype=bool)))
        if self.joint_space:
            out['true_mat'] = torch.from_numpy(np.ascontiguousarray(new_cam.camera_to_image(camera_coords), dtype=np.float32))
            out['intrinsics'] = torch.from_numpy(np.asarray(new_cam.intrinsic_matrix, dtype=np.float32))
        if self.at_test:
            out['back_rotate'] = torch.from_numpy(np.asarray(sample['camera'].R @ new_cam.R.T, dtype=np.float32))
        return out

    def __len__(self):
        return self.count if self.synthetic else len(self.samples)

    def __getitem__(self, index):
        if not self.synthetic:
            return self.parse_sample(self.samples[index])
        color, depth, cam, val = synth.make_batch(1, side=self.side_in, num_joints=self.num_joints, rank=0, step=index)
        items = [torch.from_numpy(color[0]), torch.from_numpy(cam[0]), torch.from_numpy(val[0])]
        if self.joint_space:                  # synthetic crop camera: focal 1.2 * side, principal point at the centre, joints 3 m in front of it
            side = float(self.side_in)
            intr = np.array([[1.2 * side, 0, side / 2], [0, 1.2 * side, side / 2], [0, 0, 1]], np.float32)
            placed = cam[0] + np.array([0, 0, 3000.0], np.float32)
            mat = placed[:, :2] / placed[:, 2:] * intr[[0, 1], [0, 1]] + intr[:2, 2]
            items = [items[0], torch.from_numpy(placed.astype(np.float32)), torch.from_numpy(mat.astype(np.float32)), items[2], torch.from_numpy(intr)]
        if self.at_test:
            items.append(torch.eye(3))
        return tuple(items)

"""Loader API of the reference (depth_datasets.py:23-28, datasets.py:22-27):
    data_loader(args, phase, data_info) -> torch.utils.data.DataLoader
yielding, for training, (color[B,3,S,S] f32, depth[B,1,S,S] f32, true_cam[B,J,3] f32, true_val[B,J] bool)
(depth_datasets.py:236-237) and for validation additionally back_rotate[B,3,3] (depth_datasets.py:227-229).

The real datasets need cv2 / cameralib crop re-projection and site files (out of scope).
With `-synthetic N` the loader serves N deterministic synthetic batches per epoch (synth.make_batch rules),
which is what the parity tests, smoke() and bench.py use.
"""
import numpy as np
import torch
import torch.utils.data as data

from . import synth


def data_loader(args, phase, data_info):
    dataset = Dataset(args, phase, data_info)
    return data.DataLoader(dataset, args.batch_size, shuffle=(args.shuffle and phase == 'train'), num_workers=args.workers,
                           pin_memory=True)


class Dataset(data.Dataset):

    def __init__(self, args, phase, data_info):
        assert phase in ('train', 'valid', 'test')
        if not getattr(args, 'synthetic', 0):
            raise NotImplementedError('only -synthetic N data is available: the dataset readers (cv2 + cameralib crop '
                                      're-projection, depth_datasets.py:153-237) are outside the hot-path scope')
        self.phase = phase
        self.side_in = args.side_in
        self.num_joints = args.num_joints
        self.count = args.synthetic * args.batch_size
        self.at_test = phase != 'train'
        self.do_teach = bool(getattr(args, 'do_teach', False)) and phase == 'train'
        self.stride, self.attention = args.stride, args.attention
        # -colour / -eraser: the colour stream is handed over RAW (0..255 values); augmentation + ToTensor/Normalize run on the GPU
        # in the trainer (augment.GpuAugment) instead of in the loader workers (depth_datasets.py:210)
        self.raw_color = bool(args.colour or args.eraser)

    def __len__(self):
        return self.count

    def __getitem__(self, index):
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

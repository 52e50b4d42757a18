#!/usr/bin/env python3
"""Times the GPU stage of the file-backed loaders (crops.GpuCropLoader) on synthetic NTU-sized frames: upload of a batch of raw frames,
p3d_reproject_crops (colour 1080x1920x3 uint8 with the lens model, depth 424x512 fp32 homography), p3d_enhance_depth, p3d_normalize_rgb.
Usage: python tools/crop_bench.py [--batch 64] [--side 256] [--iters 20]"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('3d-pose-estimation-with-previleged-information_amd')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--side', type=int, default=256)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--train', action='store_true', help='also time ResNet-50 training steps fed by GpuCropLoader (with / without the side-stream prefetch)')
    opt = ap.parse_args()
    rng = np.random.Generator(np.random.PCG64(0))
    cl, crops, ops = pkg.cameralib, pkg.crops, pkg.ops
    color_cam = cl.Camera([0, -3000, 1200], None, [[1050, 0, 960], [0, 1050, 540], [0, 0, 1]], [0.05, -0.08, 0.001, -0.0015, 0.01])
    color_cam.turn_towards(target_world_point=np.array([0.0, 0.0, 900.0]))
    depth_cam = cl.Camera([25, -3000, 1200], None, [[365, 0, 256], [0, 365, 212], [0, 0, 1]], None)
    depth_cam.turn_towards(target_world_point=np.array([0.0, 0.0, 900.0]))
    t0 = time.perf_counter()
    params_c, params_d = [], []
    for _ in range(opt.batch):
        box = np.array([rng.uniform(700, 900), rng.uniform(200, 300), rng.uniform(150, 300), rng.uniform(400, 600)])
        params_c.append(cl.reproject_params(color_cam, crops.plan_crop(color_cam, box, opt.side, rng.uniform(0.9, 1.1), rng.random() < 0.5)))
        dbox = box * [512 / 1920, 424 / 1080, 512 / 1920, 424 / 1080]
        params_d.append(cl.reproject_params(depth_cam, crops.plan_crop(depth_cam, dbox, opt.side, None, False)))
    plan_ms = (time.perf_counter() - t0) * 1e3 / opt.batch
    frames_c = torch.from_numpy(rng.integers(0, 256, size=(opt.batch, 1080, 1920, 3), dtype=np.uint8)).pin_memory()
    frames_d = torch.from_numpy(rng.random((opt.batch, 424, 512, 1), dtype=np.float32)).pin_memory()
    pc, pd = torch.from_numpy(np.stack(params_c)).cuda(), torch.from_numpy(np.stack(params_d)).cuda()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(opt.iters):
            fn()
        stop.record()
        torch.cuda.synchronize()
        return start.elapsed_time(stop) / opt.iters
    dev_c, dev_d = frames_c.cuda(), frames_d.cuda()
    out = dict(batch=opt.batch, side=opt.side, plan_ms_per_sample_host=round(plan_ms, 3))
    out['upload_ms'] = round(timed(lambda: (frames_c.cuda(non_blocking=True), frames_d.cuda(non_blocking=True))), 3)
    out['reproject_colour_ms'] = round(timed(lambda: ops.reproject_crops(dev_c, pc, (opt.side, opt.side), True)), 4)
    out['reproject_depth_ms'] = round(timed(lambda: ops.reproject_crops(dev_d, pd, (opt.side, opt.side), False)), 4)
    crop_c = ops.reproject_crops(dev_c, pc, (opt.side, opt.side), True)
    crop_d = ops.reproject_crops(dev_d, pd, (opt.side, opt.side), False)
    out['normalize_ms'] = round(timed(lambda: ops.normalize_rgb_(crop_c)), 4)
    out['enhance_ms'] = round(timed(lambda: ops.enhance_depth_(crop_d, 0.1, True)), 4)
    gpu_ms = out['reproject_colour_ms'] + out['reproject_depth_ms'] + out['normalize_ms'] + out['enhance_ms']
    out['gpu_stage_ms'] = round(gpu_ms, 4)
    out['crops_per_s_gpu_stage'] = round(opt.batch / gpu_ms * 1e3)
    out['crops_per_s_with_upload'] = round(opt.batch / (gpu_ms + out['upload_ms']) * 1e3)
    out['upload_GBps'] = round((frames_c.numel() + frames_d.numel() * 4) / out['upload_ms'] / 1e6, 1)
    if opt.train:
        out.update(train_fed(opt, frames_c, frames_d, np.stack(params_c), np.stack(params_d)))
    print(json.dumps(out))


def train_fed(opt, frames_c, frames_d, params_c, params_d):
    """depthnet ResNet-50 training steps (the contract workload) whose batches come through GpuCropLoader from pinned raw frames, as the DataLoader
    hands them over; decoding is not part of this (it happens in the worker processes)."""
    cam, _, tc, tv = pkg.synth.make_batch(opt.batch, side=opt.side, rank=0, step=0)[0:4]
    raw = dict(color_frame=frames_c, color_params=torch.from_numpy(params_c).pin_memory(), color_round=[True] * opt.batch,
               depth_frame=frames_d, depth_params=torch.from_numpy(params_d).pin_memory(), depth_round=[False] * opt.batch,
               depth_threshold=torch.full((opt.batch,), 0.1), nexponent=torch.ones(opt.batch, dtype=torch.bool),
               true_cam=torch.from_numpy(tc), true_val=torch.from_numpy(tv))

    class Source:
        dataset = None

        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

        def __iter__(self):
            for _ in range(self.n):
                yield raw
    args = pkg.opts.parse(['-model', 'resnet50', '-suffix', 'b', '-data_name', 'h36m', '-save_path', '/tmp/p3d', '-criterion', 'SmoothL1', '-num_joints', '17',
                           '-side_in', str(opt.side), '-batch_size', str(opt.batch)])
    model = pkg.depth_main.create_model(args)[0].cuda()
    trainer = pkg.depth_train.Trainer(args, model, pkg.utils.get_info())
    trainer.verbose = False
    res = {}
    for name, prefetch in (('inline', False), ('prefetch', True)):
        loader = pkg.crops.GpuCropLoader(Source(opt.iters + 3), opt.side, False, prefetch=prefetch)
        trainer.train(1, pkg.crops.GpuCropLoader(Source(3), opt.side, False, prefetch=prefetch))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        trainer.train(1, loader)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / (opt.iters + 3)
        res['train_ms_per_step_' + name] = round(ms, 2)
        res['train_crops_per_s_' + name] = round(opt.batch / ms * 1e3)
    return res


if __name__ == '__main__':
    main()

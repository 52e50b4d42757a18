"""Writes a miniature dataset site in the reference's on-disk layout (depth_datasets.py:95-150, datasets.py:64-72) for the loader tests:
sample / camera pickles that name the reference's class `cameralib.Camera` (as files written by its preprocessing do), split.json, colour frames
(JPEG or 8-bit PNG) and 16-bit depth PNGs, plus a metadata.json.  Everything is synthetic and seeded."""
import json
import os
import pickle
import sys
import types

import numpy as np
from PIL import Image


def _foreign_camera_class():
    """A stand-in that pickles as `cameralib.Camera` with the reference's attribute names; the package must map it onto its own class."""
    module = types.ModuleType('cameralib')

    class Camera:
        pass
    Camera.__module__, Camera.__qualname__ = 'cameralib', 'Camera'
    module.Camera = Camera
    return module, Camera


def _camera(cls, t, R, K, dist, up=(0, 0, 1)):
    cam = cls.__new__(cls)
    cam.R = np.asarray(R, np.float32)
    cam.t = np.asarray(t, np.float32)
    cam.intrinsic_matrix = np.asarray(K, np.float32)
    cam.distortion_coeffs = None if dist is None else np.asarray(dist, np.float32)
    cam.world_up = np.asarray(up)
    return cam


def _texture(rng, h, w, c):
    y, x = np.mgrid[:h, :w].astype(np.float64)
    img = np.zeros((h, w, c))
    for ch in range(c):
        for _ in range(4):
            fx, fy, ph = rng.uniform(0.01, 0.08), rng.uniform(0.01, 0.08), rng.uniform(0, 6.28)
            img[:, :, ch] += np.sin(fx * x + fy * y + ph)
    img = (img - img.min()) / (img.max() - img.min())
    return img


def _look_at(eye, target, up=(0, 0, 1)):
    z = (target - eye) / np.linalg.norm(target - eye)
    x = np.cross(z, up); x /= np.linalg.norm(x)
    return np.stack([x, np.cross(z, x), z])


def _project(world, t, R, K):
    cam = (world - t) @ R.T
    return (cam[:, :2] / cam[:, 2:]) @ K[:2, :2].T + K[:2, 2]


def make_site(root, kind, joints=17, frames=3, seed=0, color_hw=(240, 320), depth_hw=(120, 160)):
    """kind in {'ntu', 'pku', 'h36m'} -> path of the metadata.json.  Returns (metadata_path, list of sample dicts as written)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    os.makedirs(root, exist_ok=True)
    module, cls = _foreign_camera_class()
    videos = dict(ntu=['S001C001P001R001A001', 'S001C002P002R001A002', 'S002C001P003R001A003'],
                  pku=['0002-L', '0003-M', '0004-R'], h36m=['S1', 'S5', 'S9'])[kind]
    ch, cw = color_hw
    dh, dw = depth_hw
    samples, depth_cams = [], {}
    for vi, video in enumerate(videos):
        eye = rng.standard_normal(3) * 100 + [0, -3000, 1200]
        person = np.array([rng.uniform(-300, 300), rng.uniform(-200, 200), 900.0])
        Rc = _look_at(eye, person + rng.standard_normal(3) * 150)
        Kc = np.array([[cw * 0.9, 0, cw / 2 + 3], [0, cw * 0.92, ch / 2 - 2], [0, 0, 1]])
        color_dist = [0.05, -0.08, 0.001, -0.0015, 0.01] if kind == 'ntu' else None
        color_cam = _camera(cls, eye, Rc, Kc, color_dist)
        eye_d = eye + [25, 0, 0]
        Rd = _look_at(eye_d, person + rng.standard_normal(3) * 100)
        Kd = np.array([[dw * 0.95, 0, dw / 2 - 1], [0, dw * 0.95, dh / 2 + 1], [0, 0, 1]])
        depth_cam = _camera(cls, eye_d, Rd, Kd, [0.03, -0.02, 0.0005, 0.0008, 0.0] if kind == 'pku' else None)
        if kind == 'ntu':
            depth_cams[video[:8]] = depth_cam
        elif kind == 'pku':
            depth_cams[video[5]] = depth_cam
        for frame in range(frames):
            skeleton = person + rng.standard_normal((joints, 3)) * [180, 120, 420]
            valid = rng.random(joints) > 0.15
            valid[:2] = True
            pc = _project(skeleton, eye, Rc, Kc)
            lo, hi = pc.min(0) - 8, pc.max(0) + 8
            bbox = np.concatenate([lo, hi - lo])
            pd = _project(skeleton, eye_d, Rd, Kd)
            lo, hi = pd.min(0) - 4, pd.max(0) + 4
            depth_bbox = np.concatenate([lo, hi - lo])
            if kind == 'h36m':
                folder = os.path.join(root, 'images', '%s.cam%d' % (video, vi))
            else:
                folder = os.path.join(root, 'color', video)
            os.makedirs(folder, exist_ok=True)
            color = np.rint(_texture(rng, ch, cw, 3) * 255).astype(np.uint8)
            ext = '.jpg' if kind != 'pku' else '.png'
            image_path = os.path.join(folder, '%d%s' % (frame, ext))
            Image.fromarray(color).save(image_path, quality=95)
            sample = dict(video=video, frame=frame, image=image_path, camera=color_cam, bbox=bbox, skeleton=skeleton.astype(np.float32), valid=valid)
            if kind != 'h36m':
                depth = np.rint(_texture(rng, dh, dw, 1)[:, :, 0] * 2000 + 100).astype(np.uint16)
                depth[rng.random((dh, dw)) < 0.05] = 0
                if kind == 'ntu':
                    dfolder = os.path.join(root, 'nturgbd_depth_s' + video[1:4], 'nturgb+d_depth', video)
                    dpath = os.path.join(dfolder, 'Depth-' + str(frame + 1).zfill(8) + '.png')
                else:
                    dfolder = os.path.join(root, 'DEPTH_IMAGE')
                    dpath = os.path.join(dfolder, video + '.' + str(frame) + '.png')
                os.makedirs(dfolder, exist_ok=True)
                Image.fromarray(depth).save(dpath)
                sample['depth_bbox'] = depth_bbox
            samples.append(sample)
    had = sys.modules.get('cameralib')
    sys.modules['cameralib'] = module
    try:
        if kind == 'ntu':
            os.makedirs(os.path.join(root, 'final_samples'), exist_ok=True)
            half = len(samples) // 2
            for i, part in enumerate((samples[:half], samples[half:])):
                with open(os.path.join(root, 'final_samples', 'part%d.pkl' % i), 'wb') as f:
                    pickle.dump(part, f)
            with open(os.path.join(root, 'depth_cameras.pkl'), 'wb') as f:
                pickle.dump(depth_cams, f)
            split = dict(train=dict(configs=['S001C001', 'S001C002'], persons=['P001', 'P002']),
                         valid=dict(configs=['S002C001'], persons=['P003']), test=dict(configs=['S002C001'], persons=['P003']))
        elif kind == 'pku':
            with open(os.path.join(root, 'final_samples.pkl'), 'wb') as f:
                pickle.dump(samples, f)
            with open(os.path.join(root, 'cameras.pkl'), 'wb') as f:
                pickle.dump(depth_cams, f)
            split = dict(train=videos[:2], valid=videos[2:], test=videos[2:])
        else:
            with open(os.path.join(root, 'samples.pkl'), 'wb') as f:
                pickle.dump(samples, f)
            split = dict(train=videos[:2], valid=videos[2:], test=videos[2:])
    finally:
        if had is None:
            del sys.modules['cameralib']
        else:
            sys.modules['cameralib'] = had
    with open(os.path.join(root, 'split.json'), 'w') as f:
        json.dump(split, f)
    meta_path = os.path.join(root, 'metadata.json')
    with open(meta_path, 'w') as f:
        json.dump(dict(root={kind: root}, loader={kind: 'datasets' if kind == 'h36m' else 'depth_datasets'}, no_depth={kind: kind == 'h36m'},
                       thresholds={kind: dict(solid=40.0, close=80.0, rough=150.0)}), f)
    return meta_path, samples, depth_cams

"""The file-backed loaders: camera algebra against the reference's own cameralib.Camera (tests/golden/camera.npz), the host
mirror against the matrix-level oracle (oracle/np_data.py), the site readers on a miniature NTU / PKU / H36M tree (tests/site_fixture.py), and on
the GPU the batched crop re-projection, depth enhancement and the whole loader against the oracle pipeline."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import golden_path
from oracle import np_data
from site_fixture import make_site


def golden_cameras(pkg):
    g = np.load(golden_path('camera.npz'))
    for m in json.loads(str(g['meta'])):
        v = g[m['name'] + '.in']
        dist = v[21:26] if m['distorted'] else None
        yield g, m, (v[:3], v[3:12].reshape(3, 3), v[12:21].reshape(3, 3), dist, v[26:29])


def test_camera_matches_reference(pkg):
    """Every numpy-backed Camera method on the path, bit for bit against the reference class (cameralib.py:129-288, 608-620, 728-734)."""
    cl = pkg.cameralib
    for g, m, (t, R, K, dist, up) in golden_cameras(pkg):
        n = m['name']
        cam = cl.Camera(t, R, K, dist, world_up=up)
        w = g[n + '.world']
        assert np.array_equal(cam.world_to_camera(w), g[n + '.w2c'])
        assert np.array_equal(cam.world_to_image(w), g[n + '.w2i'])
        assert np.array_equal(cam.camera_to_world(cam.world_to_camera(w)), g[n + '.c2w'])
        new = cam.copy()
        new.turn_towards(target_world_point=g[n + '.target'])
        assert np.array_equal(new.R, g[n + '.R_turn'])
        new.undistort()
        new.square_pixels()
        assert np.array_equal(new.intrinsic_matrix, g[n + '.K_square'])
        far = new.world_to_image(w[:2])
        new.zoom(m['side'] / np.linalg.norm(far[0] - far[1]))
        new.center_principal_point((m['side'], m['side']))
        new.zoom(1.07)
        if m['flipped']:
            new.horizontal_flip()
        assert np.array_equal(new.intrinsic_matrix, g[n + '.K_new']) and np.array_equal(new.R, g[n + '.R_new'])
        assert np.array_equal(new.world_to_camera(w), g[n + '.new_w2c'])
        assert np.array_equal(new.camera_to_image(new.world_to_camera(w)), g[n + '.new_c2i'])
        assert np.array_equal(cam.R @ new.R.T, g[n + '.back_rotate'])
        und = cam.copy()
        und.undistort()
        assert np.array_equal(cl.get_homography(und, new), g[n + '.homography'])
        assert np.array_equal(cl.reproject_points(g[n + '.pts'], und, new), g[n + '.pts_fast'])
        # a single point is accepted wherever an array is (cameralib.py:15-30)
        assert np.array_equal(cam.world_to_image(w[3]), g[n + '.w2i'][3])


def test_oracle_camera_matches_reference(pkg):
    for g, m, (t, R, K, dist, up) in golden_cameras(pkg):
        n = m['name']
        w = g[n + '.world']
        assert np.allclose(np_data.project((w - t.astype(np.float32)).astype(np.float32) @ R.astype(np.float32).T, K, dist), g[n + '.w2i'], rtol=2e-6, atol=2e-3)
        # the oracle's crop camera with the same look-at target, box = the first two joints as its long side
        ends = g[n + '.w2i'][:2]
        K_new, R_new = np_data.crop_camera(K, R, t, dist, up, [0, 0, 1, 1], m['side'], target_world=g[n + '.target'])
        assert np.allclose(np.abs(R_new), np.abs(g[n + '.R_turn']), atol=1e-6)
        assert ends.shape == (2, 2)


def test_undistortion_round_trip(pkg):
    cl = pkg.cameralib
    for g, m, (t, R, K, dist, up) in golden_cameras(pkg):
        cam = cl.Camera(t, R, K, dist, world_up=up)
        cc = cam.world_to_camera(g[m['name'] + '.world'])
        back = cam.image_to_camera(cam.camera_to_image(cc))
        assert np.abs(back[:, :2] - cc[:, :2] / cc[:, 2:]).max() < 2e-6 and np.all(back[:, 2] == 1)
        assert np.allclose(back, np_data.unproject(cam.camera_to_image(cc), K, dist), atol=1e-6)
        assert np.allclose(cam.image_to_world(cam.camera_to_image(cc), camera_depth=2.0), cam.camera_to_world(back * 2.0), atol=1e-3)


def test_plan_crop_matches_oracle(pkg):
    """crops.plan_crop (class-based host code) against the matrix-level restatement of get_input_image, incl. -geometry zoom and the flip."""
    rng = np.random.Generator(np.random.PCG64(5))
    for g, m, (t, R, K, dist, up) in golden_cameras(pkg):
        cam = pkg.cameralib.Camera(t, R, K, dist, world_up=up)
        px = g[m['name'] + '.w2i']
        lo, hi = px.min(0), px.max(0)
        for wide in (False, True):
            size = (hi - lo) * ([1.0, 0.2] if wide else [0.2, 1.0])
            bbox = np.concatenate([(lo + hi) / 2 - size / 2, size])
            zoom, flip = (rng.uniform(0.9, 1.1), True) if wide else (None, False)
            new = pkg.crops.plan_crop(cam, bbox, 256, zoom, flip)
            K_new, R_new = np_data.crop_camera(K, R, t, dist, up, bbox, 256, zoom, flip)
            assert new.distortion_coeffs is None and np.array_equal(new.t, cam.t)
            assert np.allclose(new.R, R_new, atol=2e-6) and np.allclose(new.intrinsic_matrix, K_new, rtol=1e-5)
            # the long box side spans the crop: its end points land side_in apart (before the extra zoom), centred
            half = np.array([size[0] / 2, 0]) if size[0] >= size[1] else np.array([0, size[1] / 2])
            centre = bbox[:2] + bbox[2:] / 2
            ends = new.world_to_image(cam.image_to_world(np.stack([centre - half, centre + half])))
            assert np.linalg.norm(ends[0] - ends[1]) == pytest.approx(256 * (zoom or 1.0), rel=1e-4)
            assert np.allclose(new.world_to_image(cam.image_to_world(centre)), [128, 128], atol=1e-2)       # the box centre sits on the optical axis
            params = pkg.cameralib.reproject_params(cam, new)
            assert params.shape == (20,) and params.dtype == np.float32 and (np.any(params[15:] != 0) == (dist is not None))


def test_imread_conventions(pkg, tmp_path):
    from PIL import Image
    rgb = np.random.Generator(np.random.PCG64(0)).integers(0, 256, size=(5, 7, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(tmp_path / 'a.png')
    Image.fromarray(rgb).save(tmp_path / 'a.bmp')
    d16 = np.arange(35, dtype=np.uint16).reshape(5, 7) * 1800
    Image.fromarray(d16).save(tmp_path / 'd.png')
    png, bmp, dep = (pkg.crops.imread(str(tmp_path / n)) for n in ('a.png', 'a.bmp', 'd.png'))
    assert png.dtype == np.float32 and np.array_equal(png, rgb.astype(np.float32) / 255)
    assert bmp.dtype == np.uint8 and np.array_equal(bmp, rgb)
    assert dep.dtype == np.float32 and np.array_equal(dep, d16.astype(np.float32) / np.float32(65535))


def site_args(pkg, kind, meta, extra=()):
    flags = ['-model', 'resnet18', '-suffix', 't', '-data_name', kind, '-save_path', '/tmp/p3d', '-criterion', 'SmoothL1', '-num_joints', '17',
             '-side_in', '128', '-metadata', meta, '-batch_size', '2', '-workers', '0'] + list(extra)
    return pkg.opts.parse(flags)


@pytest.mark.parametrize('kind', ['ntu', 'pku', 'h36m'])
def test_site_readers(pkg, tmp_path, kind):
    meta, written, depth_cams = make_site(str(tmp_path / kind), kind)
    info = pkg.utils.get_info()
    mod = pkg.datasets if kind == 'h36m' else pkg.depth_datasets
    counts = {}
    for phase in ('train', 'valid'):
        ds = mod.Dataset(site_args(pkg, kind, meta), phase, info)
        counts[phase] = len(ds)
        assert all(type(s['camera']) is pkg.cameralib.Camera for s in ds.samples)          # foreign `cameralib.Camera` pickles resolved
        np.random.seed(3)
        item = ds[0]
        sample = ds.samples[0]
        np.random.seed(3)
        flip = phase == 'train' and np.random.rand() < 0.5
        cam = sample['camera']
        K_new, R_new = np_data.crop_camera(cam.intrinsic_matrix, cam.R, cam.t, cam.distortion_coeffs, cam.world_up, sample['bbox'], 128, None, flip)
        want = (np.asarray(sample['skeleton'], np.float32) - cam.t) @ R_new.T
        valid = np.asarray(sample['valid'])
        if flip:
            want, valid = want[info.mirror], valid[info.mirror]
        assert np.allclose(item['true_cam'].numpy(), want, atol=2e-2) and np.array_equal(item['true_val'].numpy(), valid)
        assert item['color_frame'].dtype == torch.uint8 and item['color_frame'].shape == (240, 320, 3)
        assert item['color_round'] == (kind != 'pku')                                       # JPEG -> rounded uint8 resampling; 8-bit PNG -> fp32 semantics
        if kind != 'h36m':
            assert item['depth_frame'].dtype == torch.float32 and item['depth_frame'].shape == (120, 160, 1) and not item['depth_round']
            assert item['depth_threshold'] == (0.1 if kind == 'ntu' else 0.5)
        if phase == 'valid':
            assert np.allclose(item['back_rotate'].numpy(), cam.R @ R_new.T, atol=1e-5)
        else:
            assert 'back_rotate' not in item
    assert counts == dict(train=6, valid=3)
    # -geometry draws its zoom from U(random_zoom, 1 / random_zoom) (depth_datasets.py:202); -do_teach adds the attention map
    if kind == 'ntu':
        ds = mod.Dataset(site_args(pkg, kind, meta, ['-geometry', '-do_teach', '-to_depth', '-teacher_path', 'x']), 'train', info)
        item = ds[1]
        assert item['atten_map'].shape[-2:] == (8, 8) and item['depth_divisor'].shape == (128, 128)
        cam = ds.depth_cam_ntu(ds.samples[1])
        assert np.allclose(item['depth_divisor'].numpy(), np.ones((128, 128), np.float32) / np_data.to_depth(np.ones((128, 128), np.float32), cam.intrinsic_matrix, None), rtol=1e-6)
    with pytest.raises(FileNotFoundError):
        mod.Dataset(pkg.opts.parse(['-model', 'resnet18', '-suffix', 't', '-data_name', kind, '-save_path', '/tmp/p3d', '-criterion', 'SmoothL1',
                                    '-num_joints', '17', '-metadata', str(tmp_path / 'missing.json')]), 'train', info)


def test_sharding_under_torchrun(pkg, tmp_path, monkeypatch):
    meta, _, _ = make_site(str(tmp_path / 'ntu'), 'ntu')
    info = pkg.utils.get_info()
    monkeypatch.setenv('WORLD_SIZE', '2')
    seen = []
    for rank in (0, 1):
        monkeypatch.setenv('RANK', str(rank))
        ds = pkg.depth_datasets.Dataset(site_args(pkg, 'ntu', meta), 'train', info)
        seen.append([(s['video'], s['frame']) for s in ds.samples])
        assert len(pkg.depth_datasets.Dataset(site_args(pkg, 'ntu', meta), 'valid', info)) == 3          # evaluation is not sharded
    assert len(seen[0]) == len(seen[1]) == 3 and not set(seen[0]) & set(seen[1])


# ------------------------------------------------------------------------------------------- GPU
def oracle_crop(frame, cam, new, round_u8):
    return np_data.reproject(frame, cam.intrinsic_matrix, cam.R, cam.distortion_coeffs, new.intrinsic_matrix, new.R, (128, 128), round_u8)


@pytest.mark.gpu
def test_reproject_crops_kernel(pkg):
    rng = np.random.Generator(np.random.PCG64(11))
    cams = list(golden_cameras(pkg))
    frames_u8 = rng.integers(0, 256, size=(len(cams), 270, 480, 3), dtype=np.uint8)
    frames_f = rng.random((len(cams), 270, 480, 1), dtype=np.float32)
    params, pairs = [], []
    for g, m, (t, R, K, dist, up) in cams:
        K = K.copy()
        K[:2] *= 0.25                                                             # the golden cameras are 1920x1080; the test frames 480x270
        cam = pkg.cameralib.Camera(t, R, K, dist, world_up=up)
        px = cam.world_to_image(g[m['name'] + '.world'])
        lo, hi = px.min(0), px.max(0)
        new = pkg.crops.plan_crop(cam, np.concatenate([lo, hi - lo]), 128, 1.05, m['flipped'])
        params.append(pkg.cameralib.reproject_params(cam, new))
        pairs.append((cam, new))
    params = torch.from_numpy(np.stack(params)).cuda()
    for frames, rounded in ((frames_u8, True), (frames_u8, False), (frames_f, False)):
        got = pkg.ops.reproject_crops(torch.from_numpy(frames).cuda(), params, (128, 128), round_u8=rounded).cpu().numpy()
        for i, (cam, new) in enumerate(pairs):
            want = oracle_crop(frames[i], cam, new, rounded)
            scale = 255.0 if frames.dtype == np.uint8 else 1.0
            diff = np.abs(got[i] - want) / scale
            # random-noise frames: a 1e-4 px difference in the sample position moves a value by up to 1e-4 of full scale; rounding can flip at .5
            assert np.mean(diff > 2e-3) < (2e-3 if rounded else 1e-6), (i, rounded, diff.max())
            assert diff.max() <= (1.0 / 255 + 1e-6 if rounded else 2e-3)
    # the single-image convenience wrapper keeps cameralib.reproject_image's output layout
    one = pkg.cameralib.reproject_image(frames_u8[1], pairs[1][0], pairs[1][1], (128, 128))
    assert one.shape == (128, 128, 3) and one.dtype == np.uint8
    assert np.mean(np.abs(one.transpose(2, 0, 1).astype(np.float32) - oracle_crop(frames_u8[1], *pairs[1], True)) > 1) < 2e-3
    with pytest.raises(pkg.ops.P3DError):
        pkg.ops.reproject_crops(torch.from_numpy(frames_u8).cuda(), params[:2], (128, 128))


@pytest.mark.gpu
def test_enhance_depth_kernel(pkg):
    rng = np.random.Generator(np.random.PCG64(12))
    x = (rng.random((3, 1, 64, 64), dtype=np.float32) * 0.2).astype(np.float32)
    x[rng.random(x.shape) < 0.1] = 0
    factor = (1 + rng.random(x.shape, dtype=np.float32)).astype(np.float32)
    for thr in (0.1, 0.5):
        for nexp in (False, True):
            for f in (None, factor):
                got = pkg.ops.enhance_depth_(torch.from_numpy(x.copy()).cuda(), thr, nexp, None if f is None else torch.from_numpy(f).cuda()).cpu().numpy()
                want = np_data.enhance(x if f is None else x / f, thr, nexp)
                edge = np.abs((x if f is None else x / f) / np.float32(10 / 255) - thr) < 1e-5       # values sitting on the threshold
                assert np.allclose(got[~edge], want[~edge], rtol=2e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize('kind,extra', [('ntu', []), ('ntu', ['-nexponent', '-to_depth', '-geometry']), ('pku', []), ('h36m', [])])
def test_loader_matches_oracle_pipeline(pkg, tmp_path, kind, extra):
    """data_loader(...) end to end on the miniature site: decoded frames -> GPU crops -> enhancement / normalisation, for the training phase
    (random flip / zoom replayed from the same numpy seed) and the validation phase, against the oracle applied sample by sample."""
    meta, _, _ = make_site(str(tmp_path / kind), kind)
    info = pkg.utils.get_info()
    mod = pkg.datasets if kind == 'h36m' else pkg.depth_datasets
    args = site_args(pkg, kind, meta, extra)
    for phase in ('train', 'valid'):
        loader = mod.data_loader(args, phase, info)
        assert len(loader) == (3 if phase == 'train' else 2)
        ds = loader.dataset
        np.random.seed(21)
        batches = list(loader)
        np.random.seed(21)
        index = 0
        for items in batches:
            color = items[0]
            assert color.is_cuda and color.dtype == torch.float32 and color.shape[1:] == (3, 128, 128)
            for b in range(color.shape[0]):
                s = ds.samples[index]
                index += 1
                flip = phase == 'train' and np.random.rand() < 0.5
                zoom = np.random.uniform(args.random_zoom, 1 / args.random_zoom)
                zoom = zoom if ('-geometry' in extra and phase == 'train') else None
                cam = s['camera']
                K_new, R_new = np_data.crop_camera(cam.intrinsic_matrix, cam.R, cam.t, cam.distortion_coeffs, cam.world_up, s['bbox'], 128, zoom, flip)
                frame = pkg.crops.imread(s['image'])
                rounded = frame.dtype == np.uint8
                frame255 = frame if rounded else frame * np.float32(255)
                want = np_data.normalize(np_data.reproject(frame255, cam.intrinsic_matrix, cam.R, cam.distortion_coeffs, K_new, R_new, (128, 128), rounded))
                diff = np.abs(color[b].cpu().numpy() - want)
                assert np.mean(diff > 0.03) < 2e-3 and diff.max() < 0.05, (phase, index, diff.max())     # 1/255/0.225 = 0.0174 per rounding flip
                if kind != 'h36m':
                    dcam = getattr(ds, 'depth_cam_' + kind)(s)
                    Kd, Rd = np_data.crop_camera(dcam.intrinsic_matrix, dcam.R, dcam.t, dcam.distortion_coeffs, dcam.world_up, s['depth_bbox'], 128, zoom, flip)
                    dframe = pkg.crops.imread(getattr(ds, 'depth_image_' + kind)(s))
                    dwant = np_data.reproject(dframe[:, :, None], dcam.intrinsic_matrix, dcam.R, dcam.distortion_coeffs, Kd, Rd, (128, 128), False)[0]
                    if '-to_depth' in extra:
                        dwant = np_data.to_depth(dwant, dcam.intrinsic_matrix, dcam.distortion_coeffs)
                    raw = dwant / np.float32(10 / 255)
                    dwant = np_data.enhance(dwant, 0.1 if kind == 'ntu' else 0.5, '-nexponent' in extra)
                    ddiff = np.abs(items[1][b, 0].cpu().numpy() - dwant)
                    edge = np.abs(raw - (0.1 if kind == 'ntu' else 0.5)) < 1e-3
                    assert ddiff[~edge].max() < 2e-3, (phase, index, ddiff[~edge].max())
                cam_i = 2 if kind != 'h36m' else 1
                want_cam = (np.asarray(s['skeleton'], np.float32) - cam.t) @ R_new.T
                valid = np.asarray(s['valid'])
                if flip:
                    want_cam, valid = want_cam[info.mirror], valid[info.mirror]
                assert np.allclose(items[cam_i][b].numpy(), want_cam, atol=2e-2) and np.array_equal(items[cam_i + 1][b].numpy(), valid)
            assert len(items) == cam_i + 2 + (phase == 'valid')
        assert index == len(ds)


@pytest.mark.gpu
def test_prefetching_loader_equals_plain_loader(pkg, tmp_path):
    """The side-stream prefetch hands out the same batches, in order, as the stage run inline on the launch stream -- also when the consumer keeps the
    GPU busy between batches (the next batch is staged while the previous one is still being used)."""
    meta, _, _ = make_site(str(tmp_path / 'pku'), 'pku')
    info = pkg.utils.get_info()
    args = site_args(pkg, 'pku', meta, ['-nexponent'])
    got = []
    for prefetch in (False, True):
        loader = pkg.depth_datasets.data_loader(args, 'valid', info)
        loader.prefetch = prefetch
        batches = []
        busy = torch.randn(2048, 2048, device='cuda')
        for items in loader:
            batches.append([t.clone() if torch.is_tensor(t) else t for t in items])
            busy = busy @ busy * 1e-3                                         # keep the launch stream occupied
        got.append(batches)
    assert len(got[0]) == len(got[1]) == 2
    for a, b in zip(*got):
        assert all(torch.equal(x, y) for x, y in zip(a, b))


@pytest.mark.gpu
def test_training_from_site_files(pkg, tmp_path):
    """One epoch of the fusion trainer and one evaluation pass fed by the file-backed loader (depth_main.main's inner loop, depth_main.py:147-160)."""
    meta, _, _ = make_site(str(tmp_path / 'ntu'), 'ntu')
    args = site_args(pkg, 'ntu', meta, ['-do_fusion', '-colour', '-eraser', '-geometry', '-shuffle'])
    info = pkg.utils.get_info()
    torch.manual_seed(0)
    model = pkg.depth_main.create_model(args)[0].cuda()
    trainer = pkg.depth_train.Trainer(args, model, info)
    trainer.verbose = False
    loader = pkg.depth_train.get_loader(args)
    train_loader = loader.data_loader(args, 'train', info)
    rec = trainer.train(1, train_loader)
    assert np.isfinite(rec['cam_train_loss']) and rec['cam_train_loss'] > 0
    test_rec = trainer.test(1, loader.data_loader(args, 'valid', info))
    assert np.isfinite(test_rec['test_loss']) and 0 <= test_rec['score_pck'] <= 1


@pytest.mark.gpu
@pytest.mark.parametrize('kind,extra', [('ntu', ['-do_fusion', '-half_acc', '-geometry', '-shuffle']), ('pku', ['-depth_only', '-partial_conv', '-nexponent']),
                                        ('h36m', [])])
def test_depth_main_on_site_files(pkg, tmp_path, kind, extra):
    """`python depth_main.py -data_name <site> ...` from the files on disk: train -> test -> checkpoint, then -val_only from that checkpoint
    (depth_main.py:111-160).  h36m goes through the RGB-only `datasets` loader metadata.json names (no_depth)."""
    meta, _, _ = make_site(str(tmp_path / kind), kind)
    flags = ['-model', 'resnet18', '-suffix', 'site', '-data_name', kind, '-save_path', str(tmp_path / 'runs'), '-criterion', 'SmoothL1', '-num_joints', '17',
             '-side_in', '128', '-batch_size', '2', '-workers', '2', '-metadata', meta] + list(extra)
    state = pkg.depth_main.main(flags + ['-n_epochs', '1', '-save_record'])
    assert state['epoch'] == 1
    root = tmp_path / 'runs' / 'resnet18-site'
    record = torch.load(root / 'train_record.pth')
    assert np.isfinite(record['cam_train_loss'][0]) and 0 <= record['score_pck'][0] <= 1
    again = pkg.depth_main.main(flags + ['-n_epochs', '1', '-val_only'])
    assert again['test_loss'] == pytest.approx(record['test_loss'][0], rel=1e-4)          # the validation phase has no random flip / zoom


@pytest.mark.gpu
@pytest.mark.parametrize('extra', [[], ['-joint_space', '-do_track']], ids=['cam', 'joint_track'])
def test_legacy_main_on_site_files(pkg, tmp_path, extra):
    """`python main.py -data_name h36m ...` from files: the RGB-only trainer, and the joint-space trainer with the least-squares track, whose loader
    tuples (true_mat, intrinsics of the crop camera) come from the same crop plan."""
    meta, _, _ = make_site(str(tmp_path / 'h36m'), 'h36m')
    flags = ['-model', 'resnet18', '-suffix', 'legacy', '-data_name', 'h36m', '-save_path', str(tmp_path / 'runs'), '-criterion', 'SmoothL1', '-num_joints', '17',
             '-side_in', '128', '-batch_size', '2', '-workers', '0', '-metadata', meta, '-n_epochs', '2', '-save_record'] + list(extra)
    if extra:
        ds = pkg.datasets.Dataset(pkg.opts.parse(flags), 'valid', pkg.utils.get_info())
        item, s = ds[0], ds.samples[0]
        new = pkg.crops.plan_crop(s['camera'], s['bbox'], 128)
        assert np.allclose(item['intrinsics'].numpy(), new.intrinsic_matrix, rtol=1e-6)
        assert np.allclose(item['true_mat'].numpy(), new.world_to_image(s['skeleton']), atol=1e-2)
        # the projected joints of a person-centred crop land around the crop
        assert np.abs(item['true_mat'].numpy() - 64).max() < 200
    state = pkg.main.main(flags)
    assert state['epoch'] == 2
    record = torch.load(tmp_path / 'runs' / 'resnet18-legacy' / 'train_record.pth')
    assert len(record['cam_train_loss']) == 2 and all(np.isfinite(record['cam_train_loss']))
    if extra:
        assert record['recon_train_loss'][0] == 0 and record['recon_train_loss'][1] > 0       # the track starts at epoch 2 (train.py:64)
        assert 0 <= record['recon_score_pck'][1] <= 1 and np.isfinite(record['mat_train_loss'][1])

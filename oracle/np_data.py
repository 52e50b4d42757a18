"""TEST INFRASTRUCTURE ONLY -- CPU (numpy) restatement of the reference's online data path, written on plain matrices (no camera class) so that
it is independent of the package's host mirror (cameralib.py / crops.py):

  * crop_camera      depth_datasets.Dataset.get_input_image camera edits            depth_datasets.py:162-191 (= datasets.py:92-117)
                     through cameralib.Camera.turn_towards / undistort / square_pixels / zoom / center_principal_point / horizontal_flip
                                                                                     cameralib.py:216-288
  * distort, project cameralib.project_points / Camera.camera_to_image               cameralib.py:129-165, 636-659
  * unproject        Camera.image_to_camera (cv::undistortPoints' 5-sweep iteration)  cameralib.py:189-201
  * reproject        cameralib.reproject_image: fast homography case and the general per-pixel case     cameralib.py:378-443, 667-711
  * enhance          depth_datasets.enhance_ntu / enhance_pku                         depth_datasets.py:39-56
  * to_depth         utils.to_depth                                                   utils.py:68-75

Parity status: the camera algebra is PINNED by tests/golden/camera.npz (the reference's own cameralib.Camera run here; cv2 mocked, so the two
cv2-backed calls -- undistortPoints and remap -- are restated from OpenCV's published definitions and are UNPINNED: cv2.remap quantises sample
positions to 1/32 px).  Only tests/ may import this module.
"""
import numpy as np

from . import np_ops


def unit(v):
    return v / np.linalg.norm(v)


def distort(plane, dist):
    k1, k2, p1, p2, k3 = [np.float32(v) for v in dist]
    x, y = plane[:, 0].astype(np.float32), plane[:, 1].astype(np.float32)
    r2 = x * x + y * y
    radial = np.float32(1) + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2
    f = radial + np.float32(2) * p2 * x + np.float32(2) * p1 * y
    return np.stack([x * f + p2 * r2, y * f + p1 * r2], 1)


def project(cam_points, K, dist=None):
    cam_points = np.asarray(cam_points, np.float32)
    plane = cam_points[:, :2] / cam_points[:, 2:]
    if dist is not None:
        plane = distort(plane, dist)
    K = np.asarray(K, np.float32)
    return plane @ K[:2, :2].T + K[:2, 2]


def unproject(pixels, K, dist=None, sweeps=5):
    K = np.asarray(K, np.float64)
    start = (np.asarray(pixels, np.float64) - K[:2, 2]) @ np.linalg.inv(K[:2, :2]).T
    if dist is None:
        plane = start
    else:
        k1, k2, p1, p2, k3 = [float(v) for v in dist]
        plane = start.copy()
        for _ in range(sweeps):
            x, y = plane[:, 0], plane[:, 1]
            r2 = x * x + y * y
            inv_radial = 1 / (1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3)
            delta = np.stack([2 * p1 * x * y + p2 * (r2 + 2 * x * x), p1 * (r2 + 2 * y * y) + 2 * p2 * x * y], 1)
            plane = (start - delta) * inv_radial[:, None]
    return np.concatenate([plane, np.ones((len(plane), 1))], 1).astype(np.float32)


def crop_camera(K, R, t, dist, world_up, bbox, side_in, zoom=None, flip=False, target_world=None):
    """-> (K_new, R_new) of the virtual crop camera; t is unchanged and the new camera has no distortion."""
    K, R, t = np.asarray(K, np.float32), np.asarray(R, np.float32), np.asarray(t, np.float32)
    bbox = np.asarray(bbox, np.float64)
    centre = bbox[:2] + bbox[2:] / 2
    ends = np.stack([centre - [bbox[2] / 2, 0], centre + [bbox[2] / 2, 0]]) if bbox[2] >= bbox[3] else \
        np.stack([centre - [0, bbox[3] / 2], centre + [0, bbox[3] / 2]])

    def to_world(pixels):
        return unproject(np.asarray(pixels, np.float32), K, dist) @ np.linalg.inv(R).T + t
    if target_world is None:
        target_world = to_world(centre[None].astype(np.float32))[0]
    z = unit(target_world - t)
    x = unit(np.cross(z, world_up))
    y = np.cross(z, x)
    R_new = np.stack([x, y, z]).astype(np.float32)
    fmean = 0.5 * (K[0, 0] + K[1, 1])
    K_new = np.array([[fmean / K[0, 0], 0, 0], [0, fmean / K[1, 1], 0], [0, 0, 1]]) @ K
    ends_new = project((to_world(ends) - t) @ R_new.T, K_new)
    K_new[:2, :2] *= side_in / np.linalg.norm(ends_new[0] - ends_new[1])
    K_new[:2, 2] = [side_in / 2, side_in / 2]
    if zoom is not None:
        K_new[:2, :2] *= zoom
    if flip:
        R_new[0] *= -1
    return K_new, R_new


def source_coords(K_old, R_old, dist_old, K_new, R_new, out_hw):
    """For every crop pixel the sampled position in the source frame (float32), [2, Ho*Wo]."""
    ho, wo = out_hw
    y, x = np.mgrid[:ho, :wo].astype(np.float32)
    grid = np.stack([x, y, np.ones_like(x)], 0).reshape(3, -1)
    if dist_old is None:                                                          # cameralib.py:672-688
        H = ((np.asarray(K_old) @ np.asarray(R_old)) @ np.linalg.inv(np.asarray(K_new) @ np.asarray(R_new))).astype(np.float32)
        c = H @ grid
        return (c[:2] / c[2:]).astype(np.float32)
    part = (np.asarray(R_old) @ np.linalg.inv(R_new) @ np.linalg.inv(K_new)).astype(np.float32)       # cameralib.py:417-423
    rays = (grid.T @ part.T).astype(np.float32)
    return project(rays, K_old, dist_old).T.astype(np.float32)


def bilinear(image_hwc, sx, sy, out_hw, round_u8):
    """Constant-border-0 bilinear sampling (the definition cv2.remap INTER_LINEAR approximates in fixed point)."""
    ho, wo = out_hw
    img = np.asarray(image_hwc)
    hs, ws = img.shape[:2]
    img = img.reshape(hs, ws, -1).astype(np.float32)
    fx, fy = np.floor(sx), np.floor(sy)
    ax, ay = (sx - fx)[:, None].astype(np.float32), (sy - fy)[:, None].astype(np.float32)
    x0, y0 = fx.astype(np.int64), fy.astype(np.int64)

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < ws) & (yy >= 0) & (yy < hs)
        return np.where(ok[:, None], img[np.clip(yy, 0, hs - 1), np.clip(xx, 0, ws - 1)], np.float32(0))
    out = (tap(y0, x0) * (1 - ax) + tap(y0, x0 + 1) * ax) * (1 - ay) + (tap(y0 + 1, x0) * (1 - ax) + tap(y0 + 1, x0 + 1) * ax) * ay
    if round_u8:
        out = np.rint(out)
    return out.reshape(ho, wo, -1).transpose(2, 0, 1).astype(np.float32)


def reproject(image_hwc, K_old, R_old, dist_old, K_new, R_new, out_hw, round_u8):
    sx, sy = source_coords(K_old, R_old, dist_old, K_new, R_new, out_hw)
    return bilinear(image_hwc, sx, sy, out_hw, round_u8)


def enhance(image01, threshold, nexponent):
    v = np.asarray(image01, np.float32) / np.float32(10.0 / 255.0)
    veil = (threshold <= v).astype(np.float32)
    return (np.exp(-v) * veil if nexponent else v / 3.0).astype(np.float32)


def to_depth(image, K, dist):
    h, w = image.shape[-2:]
    u, v = np.meshgrid(range(w), range(h))
    rays = unproject(np.stack([u, v], -1).reshape(-1, 2), K, dist).reshape(h, w, 3).astype(np.float64)
    return (image / np.sqrt(np.sum(rays ** 2, -1) + 1)).astype(np.float32)


def normalize(crop255):
    mean = np.array([0.485, 0.456, 0.406], np.float32)[:, None, None]
    dev = np.array([0.229, 0.224, 0.225], np.float32)[:, None, None]
    return ((crop255 / np.float32(255) - mean) / dev).astype(np.float32)


__all__ = ['crop_camera', 'project', 'unproject', 'reproject', 'enhance', 'to_depth', 'normalize', 'np_ops']

"""Pinhole camera with OpenCV lens distortion: the part of the reference's cameralib.py that the online data path calls
(depth_datasets.get_input_image / parse_sample, depth_datasets.py:153-237; datasets.py:83-148; utils.to_depth / transfer_bbox).

`Camera` keeps the reference's attribute names (R, t, intrinsic_matrix, distortion_coeffs, world_up; cameralib.py:44-102) so sample
and camera pickles written with the reference's class load into this one (`load_pickle`).  The geometry runs on the host in numpy -- a
few 3x3 products per sample; the per-pixel work of cameralib.reproject_image does not: `reproject_params` packs a camera pair into the
20 floats `p3d_reproject_crops` takes and the image is resampled on the GPU for a whole batch (ops.reproject_crops).

cv2 is not used: image_to_camera's undistortion restates OpenCV's published fixed-point iteration (cv::undistortPoints, 5 sweeps).
"""
import copy
import io
import pickle

import numpy as np


def _unit(v):
    return v / np.linalg.norm(v)


def _rows(points):
    points = np.asarray(points, np.float32)
    return (points[np.newaxis], True) if points.ndim == 1 else (points, False)


class Camera:

    def __init__(self, optical_center=None, rot_world_to_cam=None, intrinsic_matrix=np.eye(3), distortion_coeffs=None, world_up=(0, 0, 1),
                 extrinsic_matrix=None):
        if extrinsic_matrix is not None and (optical_center is not None or rot_world_to_cam is not None):
            raise Exception('give either `extrinsic_matrix` or `optical_center` / `rot_world_to_cam`, not both')
        if extrinsic_matrix is not None:                                        # cameralib.py:85-87
            extrinsic_matrix = np.asarray(extrinsic_matrix)
            self.R = np.asarray(extrinsic_matrix[:3, :3], np.float32)
            self.t = (-self.R.T @ extrinsic_matrix[:3, 3]).astype(np.float32)
        else:
            self.R = np.asarray(np.eye(3) if rot_world_to_cam is None else rot_world_to_cam, np.float32)
            self.t = np.asarray(np.zeros(3) if optical_center is None else optical_center, np.float32)
        self.intrinsic_matrix = np.array(intrinsic_matrix, np.float32)
        self.distortion_coeffs = None if distortion_coeffs is None else np.asarray(distortion_coeffs, np.float32)
        self.world_up = np.asarray(world_up)
        if not np.allclose(self.intrinsic_matrix[2, :], [0, 0, 1]):
            raise Exception('bottom row of the intrinsic matrix must be (0,0,1), got %s' % (self.intrinsic_matrix[2, :],))

    # ---- point transforms (cameralib.py:129-205); a single point is accepted wherever an array of points is ----
    def camera_to_image(self, points):
        points, single = _rows(points)
        if self.distortion_coeffs is not None:
            out = project_points(points, self.distortion_coeffs, self.intrinsic_matrix)
        else:
            out = (points[:, :2] / points[:, 2:]) @ self.intrinsic_matrix[:2, :2].T + self.intrinsic_matrix[:2, 2]
        return out[0] if single else out

    def world_to_camera(self, points):
        points, single = _rows(points)
        out = (points - self.t) @ self.R.T
        return out[0] if single else out

    def camera_to_world(self, points):
        points, single = _rows(points)
        out = points @ np.linalg.inv(self.R).T + self.t
        return out[0] if single else out

    def world_to_image(self, points):
        return self.camera_to_image(self.world_to_camera(points))

    def image_to_camera(self, points, depth=1):
        points, single = _rows(points)
        if self.distortion_coeffs is None:
            plane = (points - self.intrinsic_matrix[:2, 2]) @ np.linalg.inv(self.intrinsic_matrix[:2, :2]).T
        else:
            plane = undistort_to_plane(points, self.distortion_coeffs, self.intrinsic_matrix)
        out = np.concatenate([plane, np.ones_like(plane[:, :1])], axis=1).astype(np.float32) * depth
        return out[0] if single else out

    def image_to_world(self, points, camera_depth=1):
        return self.camera_to_world(self.image_to_camera(points, camera_depth))

    # ---- camera edits (cameralib.py:216-288) ----
    def zoom(self, factor):
        self.intrinsic_matrix[:2, :2] *= np.expand_dims(factor, -1)

    def scale_output(self, factor):
        self.intrinsic_matrix[:2] *= np.expand_dims(factor, -1)

    def undistort(self):
        self.distortion_coeffs = None

    def square_pixels(self):
        fx, fy = self.intrinsic_matrix[0, 0], self.intrinsic_matrix[1, 1]
        fmean = 0.5 * (fx + fy)
        self.intrinsic_matrix = np.array([[fmean / fx, 0, 0], [0, fmean / fy, 0], [0, 0, 1]]) @ self.intrinsic_matrix     # float64 from here, as in :231-238

    def horizontal_flip(self):
        self.R[0] *= -1

    def center_principal_point(self, imshape):
        self.intrinsic_matrix[:2, 2] = [imshape[1] / 2, imshape[0] / 2]

    def shift_to_center(self, desired_center_image_point, imshape):
        self.intrinsic_matrix[:2, 2] += np.float32([imshape[1], imshape[0]]) / 2 - desired_center_image_point

    def turn_towards(self, target_image_point=None, target_world_point=None):
        """Optical axis through the target, zero roll w.r.t. world_up, no flip (cameralib.py:269-288)."""
        assert (target_image_point is None) != (target_world_point is None)
        if target_image_point is not None:
            target_world_point = self.image_to_world(target_image_point)
        new_z = _unit(target_world_point - self.t)
        new_x = _unit(np.cross(new_z, self.world_up))
        new_y = np.cross(new_z, new_x)
        self.R = np.stack([new_x, new_y, new_z]).astype(np.float32)

    def get_projection_matrix(self):
        return self.intrinsic_matrix @ np.append(self.R, -self.R @ np.expand_dims(self.t, 1), axis=1)

    def copy(self):
        return copy.deepcopy(self)


def project_points(points, distortion_coeffs, intrinsic_matrix):
    """OpenCV's radial (k1 k2 k3) + tangential (p1 p2) model in fp32, coefficient order k1 k2 p1 p2 k3 (cameralib.py:636-659)."""
    k1, k2, p1, p2, k3 = (np.float32(c) for c in distortion_coeffs[:5])
    points = np.asarray(points, np.float32)
    plane = points[:, :2] / points[:, 2:]
    x, y = plane[:, 0], plane[:, 1]
    r2 = x * x + y * y
    gain = k1 * r2 + k2 * (r2 * r2) + k3 * (r2 * r2 * r2) + np.float32(1) + x * (2 * p2) + y * (2 * p1)
    bent = np.stack([x * gain + r2 * p2, y * gain + r2 * p1], axis=1)
    intrinsic_matrix = np.asarray(intrinsic_matrix, np.float32)
    return (bent @ intrinsic_matrix[:2, :2].T + intrinsic_matrix[:2, 2]).astype(np.float32)


def undistort_to_plane(points, distortion_coeffs, intrinsic_matrix, sweeps=5):
    """Pixel -> undistorted normalised image plane: cv::undistortPoints' fixed-point iteration (what cameralib.py:196-199 calls)."""
    k1, k2, p1, p2, k3 = (float(c) for c in distortion_coeffs[:5])
    k = np.asarray(intrinsic_matrix, np.float64)
    start = (np.asarray(points, np.float64) - k[:2, 2]) @ np.linalg.inv(k[:2, :2]).T
    x0, y0 = start[:, 0], start[:, 1]
    x, y = x0.copy(), y0.copy()
    for _ in range(sweeps):
        r2 = x * x + y * y
        shrink = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x, y = (x0 - dx) * shrink, (y0 - dy) * shrink
    return np.stack([x, y], axis=1).astype(np.float32)


def get_homography(src_camera, dst_camera):
    """dst-image pixel -> src-image pixel for two undistorted cameras sharing the optical centre (cameralib.py:608-620)."""
    if not np.allclose(src_camera.t, dst_camera.t):
        raise Exception('the optical centres differ: a homography cannot model this')
    return src_camera.intrinsic_matrix @ src_camera.R @ np.linalg.inv(dst_camera.R) @ np.linalg.inv(dst_camera.intrinsic_matrix)


def reproject_points(points, old_camera, new_camera):
    """Key points of an `old_camera` image in the `new_camera` image (cameralib.py:354-375, 728-734)."""
    points = np.asarray(points)
    if old_camera.distortion_coeffs is None and new_camera.distortion_coeffs is None and points.ndim == 2:
        old_matrix = old_camera.intrinsic_matrix @ old_camera.R
        new_matrix = new_camera.intrinsic_matrix @ new_camera.R
        homography = (new_matrix @ np.linalg.inv(old_matrix)).astype(np.float32)
        moved = homography[:, :2] @ points.T + homography[:, 2:]
        return (moved[:2] / moved[2:]).T
    if not np.allclose(old_camera.t, new_camera.t):
        raise Exception('the optical centre of the camera must not change, else warping is not enough')
    return new_camera.world_to_image(old_camera.image_to_world(points))


def reproject_params(old_camera, new_camera):
    """The 20 floats p3d_reproject_crops takes for one (old, new) camera pair (include/p3d_hip.h): ray[9], k[6], dist[5].

    Undistorted pair: cameralib.reproject_image_fast's homography K_old R_old (K_new R_new)^-1 in fp32 (cameralib.py:672-674) with identity k.
    Distorted old camera, undistorted new one: the per-pixel map of cameralib.py:417-425, ray = R_old R_new^-1 K_new^-1, then the old lens + K_old.
    """
    if not np.allclose(old_camera.t, new_camera.t):
        raise Exception('the optical centre of the camera must not change, else warping is not enough')
    if new_camera.distortion_coeffs is not None:
        raise NotImplementedError('re-projection INTO a distorted camera (cameralib.py:426-428) is not used by the loaders')
    out = np.zeros(20, np.float32)
    if old_camera.distortion_coeffs is None:
        old_matrix = old_camera.intrinsic_matrix @ old_camera.R
        new_matrix = new_camera.intrinsic_matrix @ new_camera.R
        out[:9] = (old_matrix @ np.linalg.inv(new_matrix)).astype(np.float32).reshape(-1)
        out[9:15] = (1, 0, 0, 0, 1, 0)
    else:
        out[:9] = (old_camera.R @ np.linalg.inv(new_camera.R) @ np.linalg.inv(new_camera.intrinsic_matrix)).astype(np.float32).reshape(-1)
        out[9:15] = np.asarray(old_camera.intrinsic_matrix, np.float32)[:2].reshape(-1)
        out[15:20] = old_camera.distortion_coeffs[:5]
    return out


def reproject_image(image, old_camera, new_camera, output_imshape, device='cuda'):
    """One image through ops.reproject_crops (the loaders batch this instead): HxW[xC] uint8 / fp32 numpy -> HoxWoxC numpy like
    cameralib.reproject_image (cameralib.py:378-443; a trailing channel axis is kept for 2-D inputs, :440-441)."""
    import torch
    from . import ops
    image = np.asarray(image)
    frame = torch.from_numpy(np.ascontiguousarray(image.reshape(image.shape[0], image.shape[1], -1)))[None].to(device)
    params = torch.from_numpy(reproject_params(old_camera, new_camera))[None].to(device)
    out = ops.reproject_crops(frame, params, tuple(output_imshape), round_u8=image.dtype == np.uint8)[0]
    out = out.permute(1, 2, 0).cpu().numpy()
    return out.astype(np.uint8) if image.dtype == np.uint8 else out


class _CameraUnpickler(pickle.Unpickler):
    """Sample / camera files written by the reference pickle its `cameralib.Camera`; resolve that name to the class above."""

    def find_class(self, module, name):
        if name == 'Camera' and module.split('.')[-1] == 'cameralib':
            return Camera
        return super().find_class(module, name)


def load_pickle(path):
    with open(path, 'rb') as file:
        return _CameraUnpickler(io.BytesIO(file.read())).load()

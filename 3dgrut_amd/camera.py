"""Camera / pose marshalling of a Batch into the C-ABI structs, without `ncore`.

Restates Tracer.__create_camera_parameters and SensorPose3DModel.__so3_matrix_to_quat
(threedgut_tracer/tracer.py:88-136, 359-488): the host inverts the C2W pose and ships the
world->sensor transform as [t(3), q(x,y,z,w)] for the start and the end of the exposure.
"""
from __future__ import annotations

import math

import numpy as np

from . import _abi

_SHUTTER = {
    "ROLLING_TOP_TO_BOTTOM": _abi.SHUTTER_ROLLING_TOP_TO_BOTTOM,
    "ROLLING_LEFT_TO_RIGHT": _abi.SHUTTER_ROLLING_LEFT_TO_RIGHT,
    "ROLLING_BOTTOM_TO_TOP": _abi.SHUTTER_ROLLING_BOTTOM_TO_TOP,
    "ROLLING_RIGHT_TO_LEFT": _abi.SHUTTER_ROLLING_RIGHT_TO_LEFT,
    "GLOBAL": _abi.SHUTTER_GLOBAL,
}
_FTHETA_POLY = {"PIXELDIST_TO_ANGLE": _abi.FTHETA_PIXELDIST_TO_ANGLE, "ANGLE_TO_PIXELDIST": _abi.FTHETA_ANGLE_TO_PIXELDIST}


_F = np.float32


def so3_matrix_to_quat_xyzw(R: np.ndarray) -> np.ndarray:
    """Unit quaternion (x,y,z,w) of a rotation matrix, in FLOAT32 and in the operation order of
    SensorPose3DModel.__so3_matrix_to_quat (tracer.py:88-136, single matrix): the reference rounds the matrix to float32 first
    (tracer.py:366-367) and runs torch float32 arithmetic on it, so the quaternion's low bits - which reach the depth keys - depend
    on that order.  Pinned bit for bit by tests/golden/pose.npz (tests/test_host_cpu.py)."""
    R = np.asarray(R, _F).reshape(3, 3)
    dm = [R[0, 0], R[1, 1], R[2, 2], (R[0, 0] + R[1, 1]) + R[2, 2]]
    c = int(np.argmax(dm))
    q = np.empty(4, _F)
    if c != 3:
        i, j, k = c, (c + 1) % 3, (c + 2) % 3
        q[i] = (_F(1) - dm[3]) + _F(2) * R[i, i]
        q[j] = R[j, i] + R[i, j]
        q[k] = R[k, i] + R[i, k]
        q[3] = R[k, j] - R[j, k]
    else:
        q[0] = R[2, 1] - R[1, 2]
        q[1] = R[0, 2] - R[2, 0]
        q[2] = R[1, 0] - R[0, 1]
        q[3] = _F(1) + dm[3]
    s = q * q
    return q / np.sqrt(((s[0] + s[1]) + s[2]) + s[3])


def world_to_sensor_tquat(c2w) -> np.ndarray:
    """[t(3), q(xyzw)] of inv(C2W) as the reference plugin derives it (tracer.py:413-419, 359-380): float64 general inverse
    (np.linalg.inv), one rounding to float32, float32 quaternion."""
    m = np.eye(4)
    m[:3, :4] = np.asarray(c2w, np.float64).reshape(-1, 4)[:3, :4]
    w2c = np.linalg.inv(m)
    return np.concatenate([w2c[:3, 3].astype(_F), so3_matrix_to_quat_xyzw(w2c[:3, :3].astype(_F))])


def _get(batch, name, default=None):
    if isinstance(batch, dict):
        return batch.get(name, default)
    return getattr(batch, name, default)


def _to_numpy(x):
    if hasattr(x, "detach"):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def _fill(arr, values):
    v = np.asarray(_to_numpy(values), np.float32).reshape(-1)
    for i in range(len(arr)):
        arr[i] = float(v[i]) if i < len(v) else 0.0


def _enum_name(v):
    return v if isinstance(v, str) else getattr(v, "name", str(v))


def camera_from_batch(gpu_batch, poses_on_device=False):
    """-> (GrutCamera, pose_start[7], pose_end[7]); raises ValueError like tracer.py:486-488.

    poses_on_device: the caller hands the camera-to-world matrices to the library as device pointers
    (GutFrame.device_T_to_world); the host-side poses are then placeholders and the pose tensors are NOT read back
    (a `.cpu()` here would wait for everything queued on the stream — the stall the reference has every iteration)."""
    if _get(gpu_batch, "rays_in_world_space", False) or poses_on_device:
        ps = pe = np.array([0, 0, 0, 0, 0, 0, 1], np.float32)
    else:
        p0 = _to_numpy(_get(gpu_batch, "T_to_world")).squeeze()
        assert p0.ndim == 2
        ps = world_to_sensor_tquat(p0)
        p1 = _get(gpu_batch, "T_to_world_end")
        if p1 is not None:
            p1 = _to_numpy(p1).squeeze()
            assert p1.ndim == 2
            pe = world_to_sensor_tquat(p1)
        else:
            pe = ps

    rays = _get(gpu_batch, "rays_ori")
    H, W = int(rays.shape[1]), int(rays.shape[2])
    cam = _abi.GrutCamera()
    cam.shutter = _abi.SHUTTER_GLOBAL
    K = _get(gpu_batch, "intrinsics")
    if K is not None:
        fx, fy, cx, cy = [float(v) for v in K]
        ow, oh = int(2 * cx), int(2 * cy)
        fovx, fovy = 2 * math.atan(ow / (2 * fx)), 2 * math.atan(oh / (2 * fy))
        cam.model = _abi.CAMERA_OPENCV_PINHOLE
        cam.width, cam.height = ow, oh
        _fill(cam.principal_point, [ow / 2.0, oh / 2.0])
        _fill(cam.focal_length, [ow / (2.0 * math.tan(fovx * 0.5)), oh / (2.0 * math.tan(fovy * 0.5))])
        return cam, ps, pe
    K = _get(gpu_batch, "intrinsics_OpenCVPinholeCameraModelParameters")
    if K is not None:
        cam.model = _abi.CAMERA_OPENCV_PINHOLE
        cam.width, cam.height = int(K["resolution"][0]), int(K["resolution"][1])
        cam.shutter = _SHUTTER[_enum_name(K["shutter_type"])]
        _fill(cam.principal_point, K["principal_point"])
        _fill(cam.focal_length, K["focal_length"])
        _fill(cam.radial, K["radial_coeffs"])
        _fill(cam.tangential, K["tangential_coeffs"])
        _fill(cam.thin_prism, K.get("thin_prism_coeffs", np.zeros(4, np.float32)))
        return cam, ps, pe
    K = _get(gpu_batch, "intrinsics_OpenCVFisheyeCameraModelParameters")
    if K is not None:
        cam.model = _abi.CAMERA_OPENCV_FISHEYE
        cam.width, cam.height = int(K["resolution"][0]), int(K["resolution"][1])
        cam.shutter = _SHUTTER[_enum_name(K["shutter_type"])]
        _fill(cam.principal_point, K["principal_point"])
        _fill(cam.focal_length, K["focal_length"])
        _fill(cam.radial, K["radial_coeffs"])
        cam.max_angle = float(K["max_angle"])
        return cam, ps, pe
    K = _get(gpu_batch, "intrinsics_FThetaCameraModelParameters")
    if K is not None:
        cam.model = _abi.CAMERA_FTHETA
        cam.width, cam.height = int(K["resolution"][0]), int(K["resolution"][1])
        cam.shutter = _SHUTTER[_enum_name(K["shutter_type"])]
        _fill(cam.principal_point, K["principal_point"])
        cam.ftheta_reference_poly = _FTHETA_POLY[_enum_name(K["reference_poly"])]
        _fill(cam.ftheta_pixeldist_to_angle, K["pixeldist_to_angle_poly"])
        _fill(cam.ftheta_angle_to_pixeldist, K["angle_to_pixeldist_poly"])
        cam.max_angle = float(K["max_angle"])
        _fill(cam.ftheta_linear_cde, K["linear_cde"])
        return cam, ps, pe
    keys = list(gpu_batch.keys()) if hasattr(gpu_batch, "keys") else [k for k in dir(gpu_batch) if not k.startswith("_")]
    raise ValueError(f"Camera intrinsics unavailable or unsupported, input keys are [{', '.join(keys)}]")

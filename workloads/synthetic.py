"""Seeded synthetic Gaussian clouds, cameras and rays — inputs for bench.py and the
parity tests.  numpy only (host side); nothing here runs in the timed region.

Cloud A "random-init" restates MixtureOfGaussians.init_from_random_point_cloud
(threedgrut/model/model.py:553-611); rays follow NeRFDataset.__get_ray_directions
(threedgrut/datasets/dataset_nerf.py:347-388): unit directions ((u+0.5-cx)/fx, (v+0.5-cy)/fy, 1),
origin 0, camera space, right-down-front C2W pose (threedgrut/datasets/protocols.py:84-96).
"""
from __future__ import annotations

import math

import numpy as np

LEGO_CAMERA_ANGLE_X = 0.6911112070083618  # NeRF-synthetic transforms_train.json


def sh_num_coeffs(degree: int) -> int:
    return (degree + 1) ** 2


def pack_density(positions, density, rotation, scale):
    """[N,12] row = {pos.xyz, density, quat.wxyz, scale.xyz, 0} (threedgut_tracer/tracer.py:176-178)."""
    n = positions.shape[0]
    out = np.zeros((n, 12), np.float32)
    out[:, 0:3] = positions
    out[:, 3] = np.reshape(density, (n,))
    out[:, 4:8] = rotation
    out[:, 8:11] = scale
    return out


def cloud_random_init(n: int, seed: int = 42, sph_degree: int = 3, default_density: float = 0.1,
                      default_scale_factor: float = 1.0):
    """Cloud A: reference random initialisation (activated parameter values)."""
    rng = np.random.default_rng(seed)
    pos = (rng.random((n, 3), dtype=np.float32) * 3.0 - 1.5).astype(np.float32)
    color = (rng.random((n, 3), dtype=np.float32) / 255.0).astype(np.float32)
    from scipy.spatial import cKDTree
    d, _ = cKDTree(pos).query(pos, k=2)
    dist = np.maximum(d[:, 1], 1e-3).astype(np.float32)
    scale = np.repeat((dist * default_scale_factor)[:, None], 3, axis=1).astype(np.float32)
    rot = rng.random((n, 4), dtype=np.float32)
    rot[:, 0] = 1.0
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    dens = np.full((n, 1), default_density, np.float32)
    sph = np.zeros((n, sh_num_coeffs(sph_degree) * 3), np.float32)
    sph[:, 0:3] = color
    return pack_density(pos, dens, rot.astype(np.float32), scale), sph


def cloud_trained_like(n: int, seed: int = 42, sph_degree: int = 3, median_scale: float = 0.01,
                       max_density: float = 0.99):
    """Cloud B: anisotropic Gaussians on a noisy unit-cube shell (realistic tile lists / early termination)."""
    rng = np.random.default_rng(seed)
    face = rng.integers(0, 6, n)
    uv = rng.random((n, 2), dtype=np.float32) * 2.0 - 1.0
    pos = np.zeros((n, 3), np.float32)
    axis = face // 2
    sign = np.where(face % 2 == 0, 1.0, -1.0).astype(np.float32)
    for a in range(3):
        m = axis == a
        o = [i for i in range(3) if i != a]
        pos[m, a] = sign[m]
        pos[m, o[0]] = uv[m, 0]
        pos[m, o[1]] = uv[m, 1]
    pos += rng.normal(0.0, 0.02, (n, 3)).astype(np.float32)
    scale = np.exp(rng.normal(math.log(median_scale), 0.7, (n, 3))).astype(np.float32)
    rot = rng.normal(0.0, 1.0, (n, 4)).astype(np.float32)
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    dens = np.clip(rng.beta(2.0, 2.0, (n, 1)), 0.005, max_density).astype(np.float32)
    sph = rng.normal(0.0, 0.3, (n, sh_num_coeffs(sph_degree) * 3)).astype(np.float32)
    return pack_density(pos, dens, rot, scale), sph


def lookat_pose(eye, target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)) -> np.ndarray:
    """C2W 4x4, camera axes right-down-front."""
    eye = np.asarray(eye, np.float64)
    f = np.asarray(target, np.float64) - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, np.asarray(up, np.float64))
    if np.linalg.norm(r) < 1e-8:
        r = np.cross(f, np.array([0.0, 1.0, 0.0]))
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = r, d, f, eye
    return m.astype(np.float32)


def orbit_pose(i: int, n_views: int = 8, radius: float = 4.0, elevation_deg: float = 30.0) -> np.ndarray:
    az = 2.0 * math.pi * (i + 0.25) / max(n_views, 1)
    el = math.radians(elevation_deg)
    eye = radius * np.array([math.cos(az) * math.cos(el), math.sin(az) * math.cos(el), math.sin(el)])
    return lookat_pose(eye)


def pinhole_intrinsics(width: int, height: int, camera_angle_x: float = LEGO_CAMERA_ANGLE_X):
    fx = fy = 0.5 * width / math.tan(0.5 * camera_angle_x)
    return [float(fx), float(fy), width / 2.0, height / 2.0]


def pinhole_rays(width: int, height: int, intrinsics):
    """camera-space rays [1,H,W,3] (origin 0, unit directions through pixel centres)."""
    fx, fy, cx, cy = intrinsics
    u, v = np.meshgrid(np.arange(width, dtype=np.float32), np.arange(height, dtype=np.float32))
    d = np.stack([(u - cx + 0.5) / fx, (v - cy + 0.5) / fy, np.ones_like(u)], -1).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    return np.zeros((1, height, width, 3), np.float32), d[None].astype(np.float32)


def fisheye_intrinsics(width: int, height: int, fov_deg: float = 160.0):
    """OpenCV-fisheye parameter dict in the Batch.intrinsics_OpenCVFisheyeCameraModelParameters layout."""
    max_angle = math.radians(fov_deg) / 2.0
    f = (min(width, height) / 2.0) / max_angle  # equidistant
    return dict(resolution=np.array([width, height], np.uint32), shutter_type="GLOBAL",
                principal_point=np.array([width / 2.0, height / 2.0], np.float32),
                focal_length=np.array([f, f], np.float32), radial_coeffs=np.zeros(4, np.float32),
                max_angle=float(max_angle))


def fisheye_rays(width: int, height: int, K):
    """camera-space unit rays of the equidistant model (zero distortion): theta = r / f."""
    cx, cy = K["principal_point"]
    fx, fy = K["focal_length"]
    u, v = np.meshgrid(np.arange(width, dtype=np.float32), np.arange(height, dtype=np.float32))
    x, y = (u + 0.5 - cx) / fx, (v + 0.5 - cy) / fy
    theta = np.sqrt(x * x + y * y)
    s = np.where(theta > 1e-8, np.sin(theta) / np.maximum(theta, 1e-8), 1.0)
    d = np.stack([x * s, y * s, np.cos(theta)], -1).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    return np.zeros((1, height, width, 3), np.float32), d[None].astype(np.float32)


def upstream_grads(width: int, height: int, seed: int = 7):
    """d_rgb, d_opacity ~ N(0,1)/P, d_dist = 0."""
    rng = np.random.default_rng(seed)
    p = width * height
    g_fd = (rng.normal(0.0, 1.0, (height, width, 4)) / p).astype(np.float32)
    g_dist = np.zeros((height, width, 1), np.float32)
    return g_fd, g_dist


class SimpleGaussians:
    """Minimal stand-in for MixtureOfGaussians' renderer-facing surface (threedgrut/model/model.py:50-118):
    activated tensors are stored directly as leaf tensors so tests / bench can read `.grad` on them."""

    def __init__(self, density12, sph, device="cuda", n_active_features=3, requires_grad=True):
        import torch
        d = torch.as_tensor(density12, dtype=torch.float32, device=device)
        self.positions = d[:, 0:3].clone().requires_grad_(requires_grad)
        self._density = d[:, 3:4].clone().requires_grad_(requires_grad)
        self._rotation = d[:, 4:8].clone().requires_grad_(requires_grad)
        self._scale = d[:, 8:11].clone().requires_grad_(requires_grad)
        self._features = torch.as_tensor(sph, dtype=torch.float32, device=device).clone().requires_grad_(requires_grad)
        self.n_active_features = n_active_features
        self.ray_feature_dim = 3

    @property
    def num_gaussians(self):
        return self.positions.shape[0]

    def get_rotation(self):
        return self._rotation

    def get_scale(self):
        return self._scale

    def get_density(self):
        return self._density

    def get_features(self):
        return self._features

    def parameters(self):
        return [self.positions, self._rotation, self._scale, self._density, self._features]

    def zero_grad(self):
        for p in self.parameters():
            p.grad = None

    def grads_packed(self):
        """(grad [N,12] in the packed layout, grad_sph [N,3*ncoef]) as numpy."""
        import torch
        n = self.num_gaussians
        z = torch.zeros((n, 1), device=self.positions.device)
        g = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.parameters()]
        packed = torch.cat([g[0], g[3], g[1], g[2], z], dim=1)
        return packed.detach().cpu().numpy(), g[4].detach().cpu().numpy()


class ActivatedGaussians:
    """Stand-in for MixtureOfGaussians with RAW parameters and the reference's activations (threedgrut/model/model.py:
    94-118, 262-274): density = sigmoid(raw), scale = exp(raw), rotation = normalize(raw), features = cat(albedo, specular).
    Used by the training-loop tests and scripts/train_synthetic.py; the renderer plugins only see the activated tensors."""

    def __init__(self, density12, sph, device="cuda", n_active_features=3):
        import torch
        d = torch.as_tensor(density12, dtype=torch.float32, device=device)
        f = torch.as_tensor(sph, dtype=torch.float32, device=device)
        self.positions = d[:, 0:3].clone().requires_grad_(True)
        dn = d[:, 3:4].clamp(1e-4, 1 - 1e-4)
        self.density = torch.log(dn / (1 - dn)).requires_grad_(True)
        self.rotation = d[:, 4:8].clone().requires_grad_(True)
        self.scale = torch.log(d[:, 8:11]).requires_grad_(True)
        self.features_albedo = f[:, :3].clone().requires_grad_(True)
        self.features_specular = f[:, 3:].clone().requires_grad_(True)
        self.n_active_features = n_active_features
        self.ray_feature_dim = 3
        self.density_activation = torch.sigmoid
        self.scale_activation = torch.exp
        self.rotation_activation = torch.nn.functional.normalize

    @property
    def num_gaussians(self):
        return self.positions.shape[0]

    def get_rotation(self):
        return self.rotation_activation(self.rotation)

    def get_scale(self):
        return self.scale_activation(self.scale)

    def get_density(self):
        return self.density_activation(self.density)

    def get_features(self):
        import torch
        return torch.cat([self.features_albedo, self.features_specular], dim=1)

    def parameters(self):
        return [self.positions, self.density, self.rotation, self.scale, self.features_albedo, self.features_specular]

    def packed(self):
        """activated [N,12] rows + SH rows as numpy (what the oracle consumes)."""
        import torch
        with torch.no_grad():
            z = torch.zeros_like(self.density)
            d12 = torch.cat([self.positions, self.get_density(), self.get_rotation(), self.get_scale(), z], dim=1)
            return d12.cpu().numpy(), self.get_features().cpu().numpy()

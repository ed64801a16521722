"""Seeded test scenes shared by the CPU (oracle) and GPU (parity) tests."""
import importlib

import numpy as np

syn = importlib.import_module("3dgrut_amd.synthetic")
camera = importlib.import_module("3dgrut_amd.camera")


def make_scene(n=2000, width=64, height=64, median_scale=0.05, seed=3, view=0, kind="trained", max_density=0.99,
               sph_degree=3):
    if kind == "trained":
        d12, sph = syn.cloud_trained_like(n, seed=seed, median_scale=median_scale, max_density=max_density, sph_degree=sph_degree)
    else:
        d12, sph = syn.cloud_random_init(n, seed=seed, sph_degree=sph_degree)
    K = syn.pinhole_intrinsics(width, height)
    ro, rd = syn.pinhole_rays(width, height, K)
    batch = dict(rays_ori=ro, rays_dir=rd, T_to_world=syn.orbit_pose(view)[None], intrinsics=K)
    cam, ps, pe = camera.camera_from_batch(batch)
    return dict(density12=d12, sph=sph, batch=batch, cam=cam, pose_start=ps, pose_end=pe, rays=(ro, rd), W=width, H=height)


def torch_batch(batch, device):
    import torch
    from types import SimpleNamespace
    return SimpleNamespace(
        rays_ori=torch.as_tensor(batch["rays_ori"], device=device),
        rays_dir=torch.as_tensor(batch["rays_dir"], device=device),
        T_to_world=torch.as_tensor(batch["T_to_world"], device=device),
        T_to_world_end=None, rays_in_world_space=False,
        intrinsics=batch.get("intrinsics"),
        intrinsics_OpenCVPinholeCameraModelParameters=batch.get("intrinsics_OpenCVPinholeCameraModelParameters"),
        intrinsics_OpenCVFisheyeCameraModelParameters=batch.get("intrinsics_OpenCVFisheyeCameraModelParameters"),
        intrinsics_FThetaCameraModelParameters=batch.get("intrinsics_FThetaCameraModelParameters"))


def rel_err(a, b):
    """||a-b||inf / (||b||inf + eps): the per-tensor gradient metric of SURVEY.md §8d."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))

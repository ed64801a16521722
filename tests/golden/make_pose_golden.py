#!/usr/bin/env python
"""Generates tests/golden/pose.npz: the sensor poses the REFERENCE plugin derives from camera-to-world matrices.

Run in the build container only (needs /root/reference):

    python tests/golden/make_pose_golden.py

The reference's `Tracer.__create_camera_parameters` (threedgut_tracer/tracer.py:385-444) and
`SensorPose3DModel.__so3_matrix_to_quat` (tracer.py:88-136) are imported from the checkout and EXECUTED (its native plugin and the
packages that are not installed here are import stubs; none of them takes part in the pose arithmetic): float32 pose -> float64
`np.linalg.inv` -> float32 rounding -> float32 torch quaternion.  The [t, q(xyzw)] pairs it returns are what 3dgrut_amd/camera.py
(host poses) and csrc/camera.hpp: c2w_to_world_to_sensor (device poses) must reproduce bit for bit.
"""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"


def import_reference_tracer():
    # packages the module imports for type annotations and enum NAMES only (not installed here / pull in the whole trainer)
    for name in ("ncore", "ncore.data", "omegaconf", "threedgrut", "threedgrut.datasets", "threedgrut.datasets.protocols"):
        sys.modules.setdefault(name, MagicMock(name=name))
    sys.path.insert(0, REFERENCE)
    import importlib
    mod = importlib.import_module("threedgut_tracer.tracer")
    mod._3dgut_plugin = MagicMock(name="lib3dgut_cc")
    return mod


def random_poses(n, seed):
    """float32 camera-to-world matrices: exact-ish rotations of every quaternion branch, orbit cameras, slightly non-rigid ones."""
    r = np.random.default_rng(seed)
    q = r.normal(size=(n, 4))
    # a quarter of the cases close to a half turn (trace ~ -1: the three diagonal branches of the quaternion extraction)
    q[: n // 4, 0] *= 1e-2
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                  2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                  2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1).reshape(n, 3, 3)
    t = r.normal(size=(n, 3)) * np.exp(r.uniform(-2, 4, (n, 1)))
    m = np.tile(np.eye(4), (n, 1, 1))
    m[:, :3, :3] = R
    m[:, :3, 3] = t
    # every eighth: a pose that is not exactly rigid (scaled / sheared by 1e-3) - the reference inverts the general matrix
    k = np.arange(n) % 8 == 7
    m[k, :3, :3] *= 1 + 1e-3 * r.normal(size=(int(k.sum()), 3, 3))
    return m.astype(np.float32)


def main():
    ref = import_reference_tracer()
    sys.path.insert(0, ROOT)
    import importlib
    syn = importlib.import_module("workloads.synthetic")
    poses = np.concatenate([random_poses(1024, 7), np.stack([syn.orbit_pose(v, n_views=8) for v in range(8)]).astype(np.float32),
                            np.eye(4, dtype=np.float32)[None]])
    ends = np.roll(poses, 1, axis=0)
    create = ref.Tracer._Tracer__create_camera_parameters
    out_s, out_e = [], []
    for c2w, c2w_end in zip(poses, ends):
        batch = types.SimpleNamespace(T_to_world=torch.from_numpy(c2w)[None], T_to_world_end=torch.from_numpy(c2w_end)[None],
                                      intrinsics=[500.0, 500.0, 320.0, 240.0], rays_in_world_space=False)
        _, pose = create(batch)
        s, e = pose.T_world_sensors
        assert s.dtype == torch.float32 and s.shape == (7,)
        out_s.append(s.numpy().copy())
        out_e.append(e.numpy().copy())
    np.savez_compressed(os.path.join(HERE, "pose.npz"), c2w=poses, c2w_end=ends, tquat_start=np.stack(out_s), tquat_end=np.stack(out_e))
    print(f"pose.npz: {len(poses)} poses")


if __name__ == "__main__":
    main()

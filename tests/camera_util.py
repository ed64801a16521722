"""Oracle renders of the orbit views used by the training surrogate tests (tests/test_optim_gpu.py)."""
import importlib

import numpy as np

import oracle

syn = importlib.import_module("workloads.synthetic")
camera = importlib.import_module("3dgrut_amd.camera")


def oracle_views(method, d12, sph, w, h, views, stride=1):
    """[views, n_rays, 3] radiance through the oracle, every `stride`-th ray of each orbit view (stride 1: all pixels)."""
    K = syn.pinhole_intrinsics(w, h)
    ro, rd = syn.pinhole_rays(w, h, K)
    out = []
    for v in range(views):
        batch = dict(rays_ori=ro, rays_dir=rd, T_to_world=syn.orbit_pose(v, n_views=views)[None], intrinsics=K)
        if method == "3dgut":
            cam, ps, pe = camera.camera_from_batch(batch)
            f = oracle.gut_forward(oracle.default_gut_config(), cam, ps, pe, 3, d12, sph, ro, rd)
            out.append(f["feat_density"][..., :3].reshape(-1, 3)[::stride])
        else:
            sel = np.arange(0, w * h, stride)
            f = oracle.grt_forward(oracle.default_grt_config(), d12, sph, 3, 1e-3, batch["T_to_world"][0], ro.reshape(-1, 3)[sel][None],
                                   rd.reshape(-1, 3)[sel][None])
            out.append(f["features"].reshape(-1, 3))
    return np.stack(out)

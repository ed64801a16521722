"""ab_identity.py [workload] — SHA-1 of every output and gradient of one forward + backward of a bench frame, for the library named by
GRUT_AMD_LIB (default: the in-tree build).  Two builds that print the same lines render the same bits (A/B of kernel variants on one box):

    python scripts/ab_identity.py c4_1m_1080p; GRUT_AMD_LIB=variants/libgrut_amd_X.so python scripts/ab_identity.py c4_1m_1080p
"""
import hashlib
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (workload table)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c4_1m_1080p"
    n, W, H, ms = bench.WORKLOADS[name]
    syn = importlib.import_module("workloads.synthetic")
    gt = importlib.import_module("3dgrut_amd.gut_tracer")
    from workloads.scenes import torch_batch
    dev = "cuda:0"
    d12, sph = syn.cloud_trained_like(n, seed=42, median_scale=ms)
    K = syn.pinhole_intrinsics(W, H)
    ro, rd = syn.pinhole_rays(W, H, K)
    batch = torch_batch(dict(rays_ori=ro, rays_dir=rd, T_to_world=syn.orbit_pose(0, n_views=8)[None], intrinsics=K), dev)
    tracer = gt.Tracer({"render": {"splat": {}}})
    g = syn.SimpleGaussians(d12, sph, device=dev)
    rng = np.random.default_rng(3)
    g_rgb = torch.as_tensor(rng.standard_normal((1, H, W, 3)).astype(np.float32) / (W * H), device=dev)
    g_opa = torch.as_tensor(rng.standard_normal((1, H, W, 1)).astype(np.float32) / (W * H), device=dev)
    out = tracer.render(g, batch, train=True)
    torch.autograd.backward([out["pred_features"], out["pred_opacity"]], [g_rgb, g_opa])
    torch.cuda.synchronize()
    gd, gs = g.grads_packed()
    items = {k: v.detach().cpu().numpy() for k, v in out.items() if torch.is_tensor(v)}
    items["grad_density"], items["grad_sph"] = np.asarray(gd), np.asarray(gs)
    print(name, os.environ.get("GRUT_AMD_LIB", "in-tree"))
    for k in sorted(items):
        print(f"  {k:24s} {hashlib.sha1(np.ascontiguousarray(items[k]).tobytes()).hexdigest()[:16]}")


if __name__ == "__main__":
    main()

import sys, os, importlib, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
from scenes import torch_batch
syn = importlib.import_module("workloads.synthetic"); gt = importlib.import_module("3dgrut_amd.gut_tracer")
n, W, H = 1_000_000, 1920, 1080
d12, sph = syn.cloud_trained_like(n, seed=42, median_scale=0.01)
K = syn.pinhole_intrinsics(W, H); ro, rd = syn.pinhole_rays(W, H, K)
batch = torch_batch(dict(rays_ori=ro, rays_dir=rd, T_to_world=syn.orbit_pose(0)[None], intrinsics=K), "cuda")
tr = gt.Tracer({"render": {"splat": {}}}); g = syn.SimpleGaussians(d12, sph)
out = tr.render(g, batch, train=True)
nat = tr.tracer_wrapper; st = nat.stats()
rng = torch.zeros((st.num_tiles, 2), dtype=torch.int32, device="cuda")
nat.lib.gut_debug_fetch(nat.handle, C.c_void_p(torch.cuda.current_stream().cuda_stream), None, None, None, None, None, None, None, C.c_void_p(rng.data_ptr()))
torch.cuda.synchronize()
r = rng.cpu().numpy().view(np.uint32).astype(np.int64); L = r[:, 1] - r[:, 0]
print("tiles", len(L), "I", L.sum(), "mean", L.mean(), "p50", np.percentile(L, 50), "p90", np.percentile(L, 90), "p99", np.percentile(L, 99), "max", L.max())
hc = out["hits_count"][0, ..., 0].detach().cpu().numpy()
print("hits/ray mean", hc.mean(), "p99", np.percentile(hc, 99), "max", hc.max())
op = out["pred_opacity"][0, ..., 0].detach().cpu().numpy()
print("opacity mean", op.mean(), "frac saturated(>0.9999)", (op > 0.9999).mean(), "frac zero", (op == 0).mean())

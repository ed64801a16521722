"""torch.profiler view of one bench step: which torch-side ops (packing, gradient routing, copies) surround the library calls."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from scenes import torch_batch  # noqa: E402

syn = importlib.import_module("workloads.synthetic")
gt = importlib.import_module("3dgrut_amd.gut_tracer")

n, W, H = 1_000_000, 1920, 1080
d12, sph = syn.cloud_trained_like(n, seed=42, median_scale=0.01)
K = syn.pinhole_intrinsics(W, H)
ro, rd = syn.pinhole_rays(W, H, K)
batch = torch_batch(dict(rays_ori=ro, rays_dir=rd, T_to_world=syn.orbit_pose(0, n_views=8)[None], intrinsics=K), "cuda")
tracer = gt.Tracer({"render": {"splat": {}}})
g = syn.SimpleGaussians(d12, sph)
g_fd = torch.as_tensor(syn.upstream_grads(W, H)[0], device="cuda")
g_rgb, g_opa = g_fd[None, ..., :3].contiguous(), g_fd[None, ..., 3:].contiguous()


def step():
    g.zero_grad()
    out = tracer.render(g, batch, train=True)
    torch.autograd.backward([out["pred_features"], out["pred_opacity"]], [g_rgb, g_opa])


for _ in range(3):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60))

"""Development aid: gradient error breakdown HIP vs f32 / f64 oracle on one scene (run on the GPU box)."""
import sys, os, importlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, oracle
from scenes import make_scene, rel_err
import test_gut_gpu as T
syn = importlib.import_module("workloads.synthetic")
jitter = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
scene = make_scene(n=3000, width=80, height=48, median_scale=0.06)
ro, rd = scene["rays"]
ro = ro + (np.random.default_rng(11).normal(size=ro.shape) * jitter).astype(np.float32)
scene["rays"] = (ro, rd); scene["batch"]["rays_ori"] = ro
g_fd, g_dist0 = syn.upstream_grads(80, 48); g_fd *= 80 * 48
g_dist = np.zeros_like(g_dist0)
gpu, ora = T._run_gpu(scene, g_fd, None), T._run_oracle(scene, g_fd, g_dist)
gd, gsph = gpu["grads"]; rd_, rsph, _ = ora["grads"]
cfg = oracle.default_gut_config()
f64 = oracle.gut_forward(cfg, scene["cam"], scene["pose_start"], scene["pose_end"], 3, scene["density12"], scene["sph"], *scene["rays"], dtype=np.float64)
r64 = oracle.gut_backward(cfg, scene["cam"], 3, f64, g_fd, g_dist, dtype=np.float64)[0]
for k, sl in {"position": slice(0,3), "density": slice(3,4), "rotation": slice(4,8), "scale": slice(8,11)}.items():
    print(f"  {k:9s} gpu-vs-f32 {rel_err(gd[:,sl], rd_[:,sl]):.3e}  gpu-vs-f64 {rel_err(gd[:,sl], r64[:,sl]):.3e}  f32-vs-f64 {rel_err(rd_[:,sl], r64[:,sl]):.3e}  max|ref| {np.abs(rd_[:,sl]).max():.3e}")
err = np.abs(gd[:, 8:11] - r64[:, 8:11]); order = np.argsort(-err.max(1))[:5]
for i in order:
    print("  particle", i, "scale", scene["density12"][i, 8:11], "gpu", gd[i, 8:11], "f32", rd_[i, 8:11], "f64", r64[i, 8:11])
cnt_diff = (gpu["out"]["hits_count"][0,...,0].detach().cpu().numpy() != ora["fwd"]["hit_count"][...,0]).sum()
print("  hit-count differing pixels", cnt_diff)

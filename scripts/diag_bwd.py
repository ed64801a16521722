import sys, os, importlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, oracle
from scenes import make_scene, rel_err
import test_gut_gpu as T
syn = importlib.import_module("3dgrut_amd.synthetic")
scene = make_scene(n=3000, width=96, height=64, median_scale=0.06)
g_fd, g_dist0 = syn.upstream_grads(96, 64); g_fd *= 96*64
for scale in (0.0, 0.1):
    g_dist = (np.random.default_rng(5).normal(size=g_dist0.shape) * scale).astype(np.float32)
    gpu, ora = T._run_gpu(scene, g_fd, g_dist), T._run_oracle(scene, g_fd, g_dist)
    ora64 = None
    gd, gsph = gpu["grads"]; rd, rsph, _ = ora["grads"]
    # f64 oracle as arbiter
    cfg = oracle.default_gut_config()
    f64 = oracle.gut_forward(cfg, scene["cam"], scene["pose_start"], scene["pose_end"], 3, scene["density12"], scene["sph"], *scene["rays"], dtype=np.float64)
    r64 = oracle.gut_backward(cfg, scene["cam"], 3, f64, g_fd, g_dist, dtype=np.float64)[0]
    print("g_dist scale", scale)
    for k, sl in {"position": slice(0,3), "density": slice(3,4), "rotation": slice(4,8), "scale": slice(8,11)}.items():
        print(f"  {k:9s} gpu-vs-f32 {rel_err(gd[:,sl], rd[:,sl]):.3e}  gpu-vs-f64 {rel_err(gd[:,sl], r64[:,sl]):.3e}  f32-vs-f64 {rel_err(rd[:,sl], r64[:,sl]):.3e}  max|ref| {np.abs(rd[:,sl]).max():.3e}")
    err = np.abs(gd[:, :3] - rd[:, :3]); i = np.unravel_index(np.argmax(err), err.shape)
    print("  worst pos entry", i, "gpu", gd[i[0], :3], "f32", rd[i[0], :3], "f64", r64[i[0], :3])
    cnt_diff = (gpu["out"]["hits_count"][0,...,0].detach().cpu().numpy() != ora["fwd"]["hit_count"][...,0]).sum()
    print("  hit-count differing pixels", cnt_diff)

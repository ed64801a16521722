"""Development aid: 3DGRT gradient error vs the oracle for the replay and the traversal backward (run on the GPU box)."""
import sys, os, importlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, oracle
from scenes import rel_err
import test_grt_gpu as T
n, w, h, scale = 6000, 64, 40, 0.05
scene = T._scene(n, w, h, scale)
rng = np.random.default_rng(4)
g_rad = rng.normal(size=(h, w, 3)).astype(np.float32); g_dns = rng.normal(size=(h, w, 1)).astype(np.float32)
cfg = oracle.default_grt_config()
ora = oracle.grt_forward(cfg, scene["density12"], scene["sph"], 3, 1e-3, scene["T"], *scene["rays"], dbg_cap=1024)
rd, rs = oracle.grt_backward(cfg, 3, 1e-3, ora, g_rad, g_dns, np.zeros_like(g_dns))
print("oracle max processed per ray", ora["hit_num"].max(), "mean", ora["hit_num"].mean())
for chunks in ("100000", "2"):
    os.environ["GRUT_GRT_LOG_CHUNKS"] = chunks
    gpu = T._render(scene, g_rad, g_dns)
    gd, gs = gpu["grads"]
    print("chunks", chunks, "pos", rel_err(gd[:, :3], rd[:, :3]), "dens", rel_err(gd[:, 3:4], rd[:, 3:4]), "rot", rel_err(gd[:, 4:8], rd[:, 4:8]),
          "scale", rel_err(gd[:, 8:11], rd[:, 8:11]), "sph", rel_err(gs, rs))

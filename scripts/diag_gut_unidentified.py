"""On the GPU box: dump the pixels of a full-size frame whose difference from the oracle identify_flips cannot explain (npz -> gpurun_out)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle, parity_util as pu

n, w, h, ms = 1_000_000, 800, 800, 0.01
inp = pu.make_frame_inputs(n, w, h, ms)
cfg = oracle.default_gut_config()
hip = pu.hip_forward(inp)
proj = oracle.gut_project(cfg, inp["cam"], inp["ps"], inp["pe"], 3, inp["d12"], inp["sph"])
proj_shared = dict(proj, rgb=hip["rgb"].astype(np.float32), tiles_count=hip["tiles_count"])
shared = oracle.gut_forward(cfg, inp["cam"], inp["ps"], inp["pe"], 3, inp["d12"], inp["sph"], *inp["rays"], proj=proj_shared,
                            lists=(hip["sorted_idx"], hip["tile_ranges"]))
d_img, d_dist = pu.pixel_errors(hip["fd"], hip["dist"], shared["feat_density"], shared["hit_distance"])
X = hip["cnt"] != shared["hit_count"][..., 0]
bad = (d_img > 1e-4) | (d_dist > 1e-4)
exempt = np.flatnonzero((X | bad).reshape(-1))
tg, _, _ = pu.identify_flips(cfg, inp["cam"], shared, proj_shared["rgb"], exempt, hip["fd"], hip["cnt"], hip["dist"])
un = exempt[tg < 0]
print("exempt", exempt.size, "unidentified", un.size)
out = {}
for k, pix in enumerate(un[:40]):
    tr = oracle.gut_pixel_trace(cfg, inp["cam"], shared, pix)
    out[f"p{k}_pix"] = pix
    out[f"p{k}_hip"] = np.concatenate([hip["fd"].reshape(-1, 4)[pix], hip["dist"].reshape(-1)[pix:pix + 1], hip["cnt"].reshape(-1)[pix:pix + 1]])
    out[f"p{k}_ora"] = np.concatenate([shared["feat_density"].reshape(-1, 4)[pix], shared["hit_distance"].reshape(-1)[pix:pix + 1], shared["hit_count"].reshape(-1)[pix:pix + 1]])
    for key in ("idx", "alpha", "hit_t", "margin"):
        out[f"p{k}_{key}"] = tr[key]
    out[f"p{k}_rgb"] = proj_shared["rgb"][tr["idx"]]
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "r02f_unidentified.npz"), **out)

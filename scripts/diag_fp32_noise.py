"""How far two fp32 evaluations of the SAME 3DGUT compositing can be apart: the HIP frame, the oracle's float build and its double
build on identical tile lists, per output (GPU box).  Answers whether BASELINE's 1e-4 absolute on the hit distance (values ~4) is
reachable by ANY fp32 evaluation: |float oracle - double oracle| is the reference arithmetic type's own rounding noise.

    python scripts/diag_fp32_noise.py [workload ...]      -> gpurun_out/fp32_noise.json
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import parity_util as pu  # noqa: E402

CASES = {"c1_100k_400": (100_000, 400, 400, 0.01), "c2_1m_800": (1_000_000, 800, 800, 0.01), "c4_1m_1080p": (1_000_000, 1920, 1080, 0.01)}


def quantiles(x):
    return {q: float(np.quantile(x, float(q))) for q in ("0.5", "0.9", "0.99", "0.999", "1.0")}


def main():
    out = {}
    for name in (sys.argv[1:] or ["c2_1m_800"]):
        n, w, h, ms = CASES[name]
        inp = pu.make_frame_inputs(n, w, h, ms)
        cfg = oracle.default_gut_config()
        hip = pu.hip_forward(inp)
        proj = oracle.gut_project(cfg, inp["cam"], inp["ps"], inp["pe"], 3, inp["d12"], inp["sph"])
        shared = dict(proj, rgb=hip["rgb"].astype(np.float32), tiles_count=hip["tiles_count"])
        lists = (hip["sorted_idx"], hip["tile_ranges"])
        o32 = oracle.gut_forward(cfg, inp["cam"], inp["ps"], inp["pe"], 3, inp["d12"], inp["sph"], *inp["rays"], proj=shared, lists=lists)
        proj64 = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) and v.dtype == np.float32 else v) for k, v in shared.items()}
        o64 = oracle.gut_forward(cfg, inp["cam"], inp["ps"], inp["pe"], 3, inp["d12"], inp["sph"], *inp["rays"], proj=proj64, lists=lists,
                                 dtype=np.float64)
        same = (hip["cnt"] == o32["hit_count"][..., 0]) & (o32["hit_count"][..., 0] == o64["hit_count"][..., 0])
        res = dict(pixels=int(same.size), pixels_same_hit_count=int(same.sum()))
        for key, a, b32, b64 in (("dist", hip["dist"][..., 0], o32["hit_distance"][..., 0], o64["hit_distance"][..., 0]),
                                 ("rgb", hip["fd"][..., :3], o32["feat_density"][..., :3], o64["feat_density"][..., :3]),
                                 ("opacity", hip["fd"][..., 3], o32["feat_density"][..., 3], o64["feat_density"][..., 3])):
            red = (lambda d: d.max(-1)) if a.ndim == 3 else (lambda d: d)
            e_h32 = red(np.abs(a - b32))[same]
            e_h64 = red(np.abs(a - b64))[same]
            e_3264 = red(np.abs(b32 - b64))[same]
            res[key] = dict(hip_vs_f32=quantiles(e_h32), hip_vs_f64=quantiles(e_h64), f32_vs_f64=quantiles(e_3264),
                            n_hip_vs_f32_gt_1e4=int((e_h32 > 1e-4).sum()), n_hip_vs_f64_gt_1e4=int((e_h64 > 1e-4).sum()),
                            n_f32_vs_f64_gt_1e4=int((e_3264 > 1e-4).sum()), max_value=float(np.abs(b64).max()))
        out[name] = res
        print(name, json.dumps(res, indent=1), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fp32_noise.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

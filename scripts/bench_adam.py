#!/usr/bin/env python
"""SelectiveAdam step on the Gaussian parameter groups of a 1M-particle model: ms per step and GB/s against the
algorithmic bytes (28 B per element of a visible row: read p, g, m, v; write p, m, v; + 4 B of flag per row)."""
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
opt_mod = importlib.import_module("3dgrut_amd.optimizers")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
widths = [3, 1, 4, 3, 3, 45]
torch.manual_seed(0)
res = {}
for frac in (1.0, 0.6, 0.1):
    params = [torch.nn.Parameter(torch.randn(n, m, device="cuda")) for m in widths]
    for p in params:
        p.grad = torch.randn_like(p)
    opt = opt_mod.SelectiveAdam([{"params": [p], "lr": 1e-3} for p in params], eps=1e-15)
    vis = (torch.rand(n, device="cuda") < frac).to(torch.int32).view(torch.float32).reshape(n, 1)
    for _ in range(3):
        opt.step(vis)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        opt.step(vis)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    nvis = int((vis.view(torch.int32) != 0).sum())
    byts = nvis * sum(widths) * 28 + n * 4
    res[f"visible_{frac}"] = {"ms": ms, "GB/s": byts / ms / 1e6, "frac_of_8TBs": byts / ms / 1e6 / 8000.0}
print(json.dumps({"selective_adam": {"rows": n, "row_floats": sum(widths), **res}}))

"""Single-GPU cost of the local half of the factored gradient exchange (3dgrut_amd/dp.py): grut_sph_grad_from_views for V gathered
view factors, next to the plain expansion it replaces.  The collectives themselves need a multi-GPU node."""
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
abi = importlib.import_module("3dgrut_amd._abi")


def time_ms(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


out = []
for n in (1_000_000, 3_000_000):
    pos = torch.randn(n, 12, device="cuda")
    for v in (1, 2, 4, 8):
        f = torch.randn(v, n + 1, 3, device="cuda")
        f[:, :n][torch.rand(v, n, device="cuda") < 0.34] = 0.0        # a third of the particles invisible per view
        f[:, n] = torch.randn(v, 3, device="cuda") * 4.0
        ms = time_ms(lambda: abi.sph_grad_from_views(f, pos, 3, 3))
        moved = (v * (n + 1) * 12 + n * 12 + n * 192) / 1e9
        out.append(dict(particles=n, views=v, ms=round(ms, 4), gbytes=round(moved, 3), tb_per_s=round(moved / ms, 2)))
print(json.dumps(out))

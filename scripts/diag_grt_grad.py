"""On the GPU box: which ray makes the 3DGRT gradient of the worst particle differ from the oracle's (c3_grt_100k_400)?
Per-ray backward on both sides for every ray that processed that particle."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import parity_util as pu  # noqa: E402
from scenes import torch_batch  # noqa: E402
import torch  # noqa: E402

syn = importlib.import_module("workloads.synthetic")
grt = importlib.import_module("3dgrut_amd.grt_tracer")
n, w, h, cap = 100_000, 400, 400, 192
inp = pu.make_frame_inputs(n, w, h, 0.01)
d12, sph = inp["d12"], inp["sph"]
tr = grt.Tracer({"render": {"enable_hitcounts": True}})
g = syn.SimpleGaussians(d12, sph)
tr.build_acc(g, rebuild=True)
nat = tr.tracer_wrapper
batch = torch_batch(inp["batch"], "cuda")
frame = nat.make_frame(0, 3, tr._min_transmittance, n, h, w, batch.T_to_world)
d12_t, sph_t = torch.as_tensor(d12, device="cuda").contiguous(), torch.as_tensor(sph, device="cuda").contiguous()
feat, dns, hit, nrm, cnt, vis, ids, num = nat.trace(frame, d12_t, sph_t, batch.rays_ori.contiguous(), batch.rays_dir.contiguous(), hit_capacity=cap)
inst = nat.instances(n, "cuda").cpu().numpy()
scene_aabb = np.array(list(nat.stats().scene_aabb), np.float32)
ids, num, cnt = ids.cpu().numpy().view(np.uint32), num.cpu().numpy().astype(np.int64), cnt[0].cpu().numpy().reshape(-1)
cfg = oracle.default_grt_config()
ro, rd = inp["rays"]
T = inp["batch"]["T_to_world"][0]
ora = oracle.grt_forward(cfg, d12, sph, 3, tr._min_transmittance, T, ro.reshape(1, -1, 3), rd.reshape(1, -1, 3), inst=inst, scene=scene_aabb, dbg_cap=cap)
F = (num != ora["hit_num"]) | (cnt != ora["hit_count"].reshape(-1))
rng = np.random.default_rng(4)
g_rad = rng.normal(size=(h, w, 3)).astype(np.float32)
g_dns = rng.normal(size=(h, w, 1)).astype(np.float32)
g_rad[F.reshape(h, w)] = 0
g_dns[F.reshape(h, w)] = 0


def hip_grad(gr, gdn):
    g.zero_grad()
    out = tr.render(g, batch, train=True)
    loss = (out["pred_features"][0] * torch.as_tensor(gr, device="cuda")).sum() + (out["pred_opacity"][0] * torch.as_tensor(gdn, device="cuda")).sum()
    loss.backward()
    torch.cuda.synchronize()
    return g.grads_packed()


def ora_grad(sel, gr, gdn):
    sub = dict(ora)
    for k in ("features", "density", "hit_distance"):
        sub[k] = ora[k].reshape(-1, ora[k].shape[-1])[sel][None]
    sub["rays"] = (ro.reshape(-1, 3)[sel][None], rd.reshape(-1, 3)[sel][None])
    return oracle.grt_backward(cfg, 3, tr._min_transmittance, sub, gr.reshape(-1, 3)[sel][None], gdn.reshape(-1, 1)[sel][None], np.zeros((1, sel.size, 1), np.float32))


gd, gs = hip_grad(g_rad, g_dns)
allr = np.arange(w * h)
rdg, rsg = ora_grad(allr, g_rad, g_dns)
err = np.abs(gd[:, :11] - rdg[:, :11])
scale = np.abs(rdg[:, :11]).max(0)
for name, sl in pu.GRAD_SLICES.items():
    print(name, "rel err", float(err[:, sl].max() / np.abs(rdg[:, sl]).max()))
worst = int(np.argmax((err / scale).max(1)))
print("worst particle", worst, "hip", gd[worst, :11], "oracle", rdg[worst, :11], "density12", d12[worst])
# the same backward through the traversal fallback instead of the hit-log replay
os.environ["GRUT_GRT_LOG_CHUNKS"] = "3"
gd_t, gs_t = hip_grad(g_rad, g_dns)
del os.environ["GRUT_GRT_LOG_CHUNKS"]
err_t = np.abs(gd_t[:, :11] - rdg[:, :11])
for name, sl in pu.GRAD_SLICES.items():
    print(name, "rel err (traversal backward)", float(err_t[:, sl].max() / np.abs(rdg[:, sl]).max()), " replay vs traversal", float(np.abs(gd - gd_t)[:, sl].max() / np.abs(rdg[:, sl]).max()))
print("worst particle, traversal backward:", gd_t[worst, :11])
gd2, _ = hip_grad(g_rad, g_dns)
print("replay run-to-run:", float((np.abs(gd2 - gd)[:, :11] / scale).max()))
rays = np.flatnonzero((ids == worst).any(1))
print("rays that processed it:", rays.size)
rows = []
for r in rays:
    gr, gdn = np.zeros_like(g_rad), np.zeros_like(g_dns)
    gr.reshape(-1, 3)[r] = g_rad.reshape(-1, 3)[r]
    gdn.reshape(-1, 1)[r] = g_dns.reshape(-1, 1)[r]
    a, _ = hip_grad(gr, gdn)
    b, _ = ora_grad(np.array([r]), gr, gdn)
    e = float((np.abs(a[worst, :11] - b[worst, :11]) / scale).max())
    rows.append((e, int(r)))
tot_h = np.zeros(11); tot_o = np.zeros(11)
for r in rays:
    gr, gdn = np.zeros_like(g_rad), np.zeros_like(g_dns)
    gr.reshape(-1, 3)[r] = g_rad.reshape(-1, 3)[r]
    gdn.reshape(-1, 1)[r] = g_dns.reshape(-1, 1)[r]
    tot_h += hip_grad(gr, gdn)[0][worst, :11]
    tot_o += ora_grad(np.array([r]), gr, gdn)[0][worst, :11]
print("sum of per-ray hip   :", tot_h)
print("sum of per-ray oracle:", tot_o)
# wave (8x8 pixel block) of each ray, and how many rays of the list share a block
bx, by = (rays % w) // 8, (rays // w) // 8
blk, cnts = np.unique(by * ((w + 7) // 8) + bx, return_counts=True)
print("blocks:", dict(zip(blk.tolist(), cnts.tolist())))
# per-block totals: all rays of one 8x8 block at once (what one wave aggregates) against the oracle
for b in blk:
    sel = rays[(by * ((w + 7) // 8) + bx) == b]
    gr, gdn = np.zeros_like(g_rad), np.zeros_like(g_dns)
    gr.reshape(-1, 3)[sel] = g_rad.reshape(-1, 3)[sel]
    gdn.reshape(-1, 1)[sel] = g_dns.reshape(-1, 1)[sel]
    a = hip_grad(gr, gdn)[0][worst, :11]
    o = ora_grad(sel, gr, gdn)[0][worst, :11]
    print(f"  block {b}: {sel.size} rays, slots {[int(np.flatnonzero(ids[r, :num[r]] == worst)[0]) for r in sel]}, err {float((np.abs(a - o) / scale).max()):.3e}")
rows.sort(reverse=True)
print("per-ray error of that particle's gradient (relative to the tensor maxima), worst first:", rows[:6])
e, r = rows[0]
k = int(num[r])
pos = int(np.flatnonzero(ids[r, :k] == worst)[0])
print(f"ray {r}: processed {k} hits (oracle {int(ora['hit_num'][r])}), particle at position {pos}; hit count {cnt[r]} / {ora['hit_count'].reshape(-1)[r]}; flip-masked: {bool(F[r])}")
print("  hip out", feat[0].reshape(-1, 3)[r].cpu().numpy(), dns[0].reshape(-1)[r].item(), hit[0].reshape(-1, 2)[r].cpu().numpy())
print("  ora out", ora["features"].reshape(-1, 3)[r], ora["density"].reshape(-1)[r], ora["hit_distance"].reshape(-1, 2)[r])
gr, gdn = np.zeros_like(g_rad), np.zeros_like(g_dns)
gr.reshape(-1, 3)[r] = g_rad.reshape(-1, 3)[r]
gdn.reshape(-1, 1)[r] = g_dns.reshape(-1, 1)[r]
a, _ = hip_grad(gr, gdn)
b, _ = ora_grad(np.array([r]), gr, gdn)
touched = np.flatnonzero(np.abs(b[:, :11]).max(1) > 0)
touched_h = np.flatnonzero(np.abs(a[:, :11]).max(1) > 0)
print("  particles with gradient from this ray: oracle", touched.size, "hip", touched_h.size, "symmetric difference", np.setxor1d(touched, touched_h))
pe = (np.abs(a[:, :11] - b[:, :11]) / scale).max(1)
bad = np.argsort(pe)[-5:][::-1]
for p in bad:
    where = np.flatnonzero(ids[r, :k] == p)
    print(f"  particle {p} at hit position {where}: err {pe[p]:.3e} hip {a[p, :11]} oracle {b[p, :11]}")

"""On the GPU box: lifetime distribution of the forward sweep's waves on the bench frame (instrumented kernel): how much of the kernel's
duration is tail (few long-running waves) and how busy the SIMD slots are."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from scenes import torch_batch
syn = importlib.import_module("workloads.synthetic"); gt = importlib.import_module("3dgrut_amd.gut_tracer"); abi = importlib.import_module("3dgrut_amd._abi")
n, W, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1_000_000, 1920, 1080)
d12, sph = syn.cloud_trained_like(n, seed=42, median_scale=0.01)
K = syn.pinhole_intrinsics(W, H); ro, rd = syn.pinhole_rays(W, H, K)
batch = torch_batch(dict(rays_ori=ro, rays_dir=rd, T_to_world=syn.orbit_pose(0, n_views=8)[None], intrinsics=K), "cuda")
tr = gt.Tracer({"render": {"splat": {}}}); nat = tr.tracer_wrapper
g = syn.SimpleGaussians(d12, sph)
for _ in range(3):
    tr.render(g, batch, train=True)
abi.check(nat.lib.gut_profile_enable(nat.handle, 2), "enable")
out = tr.render(g, batch, train=True)
g_fd = torch.as_tensor(syn.upstream_grads(W, H)[0], device="cuda")
torch.autograd.backward([out["pred_features"], out["pred_opacity"]], [g_fd[None, ..., :3].contiguous(), g_fd[None, ..., 3:].contiguous()])
torch.cuda.synchronize()
st = nat.stats()
nb = ((int(st.num_tiles) + 7) // 8 * 8) * 2
nbwd = (int(st.num_intersections) // 256 + 1 + 7) // 8 * 8 * 2 + nb + 64
buf = torch.zeros(16 + 4 * nb + 2 * nbwd, dtype=torch.int64, device="cuda")
abi.check(nat.lib.gut_debug_fetch_work(nat.handle, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(buf.data_ptr()), buf.numel()), "fetch")
torch.cuda.synchronize()
raw = buf.cpu().numpy()
wb = raw[16 + 4 * nb:].reshape(-1, 2)
w = raw[16:16 + 4 * nb].reshape(-1, 4)
life_all, start_all = (w[:, 0] & 0xFFFFFFFF).astype(np.float64), w[:, 1].astype(np.float64)
t_rays_all = ((w[:, 0] >> 32) & 0xFFFF).astype(np.float64) * 0.01
t_stage_all = ((w[:, 0] >> 48) & 0xFFFF).astype(np.float64) * 0.01
ev_all, acc_all, len_all = (w[:, 2] & 0xFFFFFFFF).astype(np.float64), (w[:, 2] >> 32).astype(np.float64), (w[:, 3] & 0xFFFFFF).astype(np.float64)
rounds_all = ((w[:, 3] >> 24) & 1023).astype(np.float64)
t_first_all, t_done_all = ((w[:, 3] >> 34) & 32767).astype(np.float64) * 0.01, ((w[:, 3] >> 49) & 32767).astype(np.float64) * 0.01
print("fwd evaluated", int(st.fwd_entries_evaluated), "accepted", int(st.fwd_entries_accepted))
# wall_clock64: one 100 MHz counter for the chip
life, start = life_all, start_all
ok = life > 0
life, start = life[ok] * 0.01, start[ok] * 0.01   # microseconds
t0, t1 = start.min(), (start + life).max()
span = t1 - t0
print(f"waves {ok.sum()} kernel span {span:.1f} us; wave lifetime mean {life.mean():.1f} p50 {np.percentile(life, 50):.1f} p90 {np.percentile(life, 90):.1f} "
      f"p99 {np.percentile(life, 99):.1f} max {life.max():.1f} us")
print("sum of lifetimes / (span x 6144 slots) = %.3f" % (life.sum() / (span * 6144)))
edges = np.linspace(t0, t1, 21)
occ = [(np.minimum(start + life, edges[i + 1]) - np.maximum(start, edges[i])).clip(min=0).sum() / ((edges[i + 1] - edges[i]) * 6144) for i in range(20)]
print("slot occupancy over the kernel's duration (20 bins):", " ".join("%.2f" % o for o in occ))
ends = np.sort(start + life)
print("fraction of the span when 50 / 90 / 99 %% of the waves have finished: %.2f %.2f %.2f" % tuple((np.percentile(ends, [50, 90, 99]) - t0) / span))
ev, ac, ln = ev_all[ok], acc_all[ok], len_all[ok]
print("correlation of wave lifetime with: evaluated %.3f, accepted %.3f, list length %.3f" % (np.corrcoef(life, ev)[0, 1], np.corrcoef(life, ac)[0, 1], np.corrcoef(life, ln)[0, 1]))
A = np.stack([ev - ac, ac, np.ones_like(ev)], 1)
coef, *_ = np.linalg.lstsq(A, life, rcond=None)
print("lifetime ~ %.3f us x rejected + %.3f us x accepted + %.1f us (least squares); residual std %.1f us" % (coef[0], coef[1], coef[2], (life - A @ coef).std()))
rd = rounds_all[ok]
A = np.stack([ev - ac, ac, rd, np.ones_like(ev)], 1)
coef, *_ = np.linalg.lstsq(A, life, rcond=None)
print("lifetime ~ %.3f us x rejected + %.3f us x accepted + %.2f us x staged rounds + %.1f us; residual std %.1f us; rounds per wave mean %.1f, staged entries evaluated %.2f" % (
    coef[0], coef[1], coef[2], coef[3], (life - A @ coef).std(), rd.mean(), ev.sum() / max(1.0, 64 * rd.sum())))
for lo, hi in ((0, 1), (1, 16), (16, 64), (64, 128), (128, 192), (192, 256), (256, 1 << 30)):
    sel = (ev >= lo) & (ev < hi)
    if sel.any():
        print("  waves with %4d <= evaluated < %-6d: %6d waves, lifetime mean %6.1f p10 %6.1f p90 %6.1f us, list length mean %6.0f, accepted mean %5.1f" % (
            lo, min(hi, 99999), sel.sum(), life[sel].mean(), *np.percentile(life[sel], [10, 90]), ln[sel].mean(), ac[sel].mean()))
tr, tf, td = t_rays_all[ok], t_first_all[ok], t_done_all[ok]
print("phases (us, mean / p50 / p90): rays + range ready %.1f / %.1f / %.1f; first round staged (from there) %.1f / %.1f / %.1f; rest of the sweep %.1f; after the sweep (stores, exit) %.1f / %.1f / %.1f" % (
    tr.mean(), *np.percentile(tr, [50, 90]), tf.mean(), *np.percentile(tf, [50, 90]), (td - tf).mean(), (life - tr - td).mean(), *np.percentile(life - tr - td, [50, 90])))
ts = t_stage_all[ok]
print("time in staging (wait for the round's loads + record + cull + barrier), all rounds: mean %.1f us per wave = %.2f us per round; compositing loops: %.1f us per wave" % (
    ts.mean(), ts.sum() / max(1.0, rd.sum()), (td - ts).mean()))
A = np.stack([ev - ac, ac, np.ones_like(ev)], 1)
coef, *_ = np.linalg.lstsq(A, td - ts, rcond=None)
print("compositing loops ~ %.3f us x rejected + %.3f us x accepted + %.1f us; residual std %.1f us" % (coef[0], coef[1], coef[2], ((td - ts) - A @ coef).std()))
late = np.argsort(start + life)[-160:]
print("the last 1 %% of the waves to finish: evaluated mean %.0f (all: %.0f), accepted mean %.0f (all: %.0f), list length mean %.0f (all: %.0f), "
      "lifetime mean %.1f us, start mean at %.2f of the span" % (ev[late].mean(), ev.mean(), ac[late].mean(), ac.mean(), ln[late].mean(), ln.mean(), life[late].mean(),
                                                              ((start[late] - t0) / span).mean()))
order = np.argsort(-ev)
print("evaluated entries per wave: p50 %.0f p90 %.0f p99 %.0f max %.0f; share of all evaluated entries in the top 1 %% / 5 %% waves: %.3f / %.3f" % (
    *np.percentile(ev, [50, 90, 99]), ev.max(), ev[order[:163]].sum() / ev.sum(), ev[order[:816]].sum() / ev.sum()))
# ---- gradient sweep ----
lb, sb = wb[:, 0].astype(np.float64) * 0.01, wb[:, 1].astype(np.float64) * 0.01
okb = lb > 0
lb, sb = lb[okb], sb[okb]
b0, b1 = sb.min(), (sb + lb).max()
bspan = b1 - b0
print(f"gradient sweep: tasks run {okb.sum()}, span {bspan:.1f} us, task lifetime mean {lb.mean():.1f} p50 {np.percentile(lb, 50):.1f} p90 {np.percentile(lb, 90):.1f} "
      f"p99 {np.percentile(lb, 99):.1f} max {lb.max():.1f} us; busy {lb.sum() / (bspan * 4096):.3f} of 4096 slots")
edges = np.linspace(b0, b1, 21)
occ = [(np.minimum(sb + lb, edges[i + 1]) - np.maximum(sb, edges[i])).clip(min=0).sum() / ((edges[i + 1] - edges[i]) * 4096) for i in range(20)]
print("slot occupancy (20 bins):", " ".join("%.2f" % o for o in occ))

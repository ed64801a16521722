"""On the GPU box: per-packet records of the 3DGRT training forward on the c3_grt_1m_800 frame (instrumented launch, GRUT_GRT_COUNT=1) -
start, lifetime, wave-level tests and list length of every 8x8 ray packet, saved to gpurun_out/<tag>_packets.npz for offline analysis
(which static quantity predicts a packet's lifetime, what an order by it would give), plus the forward's time with and without the hit log.

    python scripts/diag_grt_packets.py [tag]
"""
import ctypes as C, heapq, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from workloads.scenes import torch_batch
syn = importlib.import_module("workloads.synthetic"); grt = importlib.import_module("3dgrut_amd.grt_tracer"); abi = importlib.import_module("3dgrut_amd._abi")
tag = sys.argv[1] if len(sys.argv) > 1 else "diag"
n, W, H = 1_000_000, 800, 800
d12, sph = syn.cloud_trained_like(n, seed=42, median_scale=0.01)
K = syn.pinhole_intrinsics(W, H); ro, rd = syn.pinhole_rays(W, H, K)
batch = torch_batch(dict(rays_ori=ro, rays_dir=rd, T_to_world=syn.orbit_pose(0, n_views=8)[None], intrinsics=K), "cuda")
tr = grt.Tracer({"render": {"enable_kernel_timings": True}}); nat = tr.tracer_wrapper
g = syn.SimpleGaussians(d12, sph, device="cuda")
tr.build_acc(g, rebuild=True)


def fwd_ms(train, reps=5):
    ts = []
    for _ in range(reps + 2):
        if train:
            out = tr.render(g, batch, train=True)
        else:
            with torch.no_grad():
                out = tr.render(g, batch, train=False)
        torch.cuda.synchronize()
        ts.append(tr.timings["forward_render"])
    return float(np.median(ts[2:]))


print(f"forward_render: training (hit log) {fwd_ms(True):.3f} ms, inference (no log) {fwd_ms(False):.3f} ms")
os.environ["GRUT_GRT_COUNT"] = "1"
out = tr.render(g, batch, train=True)
torch.cuda.synchronize()
del os.environ["GRUT_GRT_COUNT"]
gx, gy = (W + 7) // 8, (H + 7) // 8
ST = 8
sx, sy = (gx + ST - 1) // ST, (gy + ST - 1) // ST
nst = (sx * sy + 7) & ~7
nblk = nst * ST * ST
nrec = sx * sy * ST * ST + 512   # (the size grt_forward gives the instrumented launch's record table)
buf = torch.zeros(16 + 3 * nrec, dtype=torch.int64, device="cuda")
abi.check(nat.lib.grt_debug_fetch_work(nat.handle, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(buf.data_ptr()), buf.numel()), "fetch")
torch.cuda.synchronize()
raw = buf.cpu().numpy()
w = np.zeros((nblk, 3), dtype=np.int64); w[:min(nblk, nrec)] = raw[16:16 + 3 * min(nblk, nrec)].reshape(-1, 3)
b = np.arange(nblk)
xcd, idx = b & 7, b >> 3
st, within = (idx // (ST * ST)) * 8 + xcd, idx % (ST * ST)
bx, by = (st % sx) * ST + within % ST, (st // sx) * ST + within // ST
inside = (bx < gx) & (by < gy) & (w[:, 1] > 0)
ranges, entries = nat.fetch_lists(W, H, "cuda")
ranges = ranges.cpu().numpy().astype(np.int64)
pidx = by * gx + bx
llen = np.zeros(nblk); llen[inside] = (ranges[pidx[inside], 1] - ranges[pidx[inside], 0])
start, life = w[:, 0].astype(np.float64) * 0.01, w[:, 1].astype(np.float64) * 0.01   # us
tests = (w[:, 2] & 0xFFFFFFFF).astype(np.float64)
hits = out["hits_count"][0, :, :, 0].detach().cpu().numpy() if "hits_count" in out else None
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"{tag}_packets.npz"), bx=bx, by=by, inside=inside, start=start, life=life, tests=tests, llen=llen,
                    opacity=out["pred_opacity"][0, :, :, 0].detach().cpu().numpy(), hits=hits if hits is not None else np.zeros(1))
s, l, t, ll = start[inside], life[inside], tests[inside], llen[inside]
t0, t1 = s.min(), (s + l).max(); span = t1 - t0
print(f"packets {inside.sum()}, span {span / 1e3:.2f} ms; lifetime mean {l.mean() / 1e3:.3f} p50 {np.percentile(l, 50) / 1e3:.3f} p90 {np.percentile(l, 90) / 1e3:.3f} "
      f"p99 {np.percentile(l, 99) / 1e3:.3f} max {l.max() / 1e3:.3f} ms; sum / (span x 4096) = {l.sum() / (span * 4096):.3f}")
edges = np.linspace(t0, t1, 21)
print("slot occupancy (20 bins):", " ".join("%.2f" % ((np.minimum(s + l, edges[i + 1]) - np.maximum(s, edges[i])).clip(min=0).sum() / ((edges[i + 1] - edges[i]) * 4096)) for i in range(20)))
print("correlation of lifetime with: wave tests %.3f, list length %.3f" % (np.corrcoef(t, l)[0, 1], np.corrcoef(ll, l)[0, 1]))
cx, cy = (bx[inside] + 0.5) / gx - 0.5, (by[inside] + 0.5) / gy - 0.5
rr = np.sqrt(cx * cx + cy * cy)
print("lifetime by distance from the image centre (deciles of r): " + " ".join("%.2f" % (l[(rr >= np.percentile(rr, 10 * i)) & (rr <= np.percentile(rr, 10 * i + 10))].mean() / 1e3) for i in range(10)))
for name, order in (("launch order", np.argsort(s)), ("longest first", np.argsort(-l)), ("most tests first", np.argsort(-t)), ("longest list first", np.argsort(-ll)), ("centre first", np.argsort(rr))):
    slots = [0.0] * 4096
    heapq.heapify(slots)
    for i in order:
        x = heapq.heappop(slots)
        heapq.heappush(slots, x + l[i])
    print(f"greedy schedule of the measured lifetimes, {name}: makespan {max(slots) / 1e3:.2f} ms")
names = ["wave_node_visits", "lane_tests", "processed_hits", "lane_rounds", "lane_inserts", "lane_passed_all", "lane_rej_t_range", "lane_rej_box", "lane_rej_distance",
         "wave_tests", "wave_tests_with_a_lane_in_t_range", "wave_tests_with_an_insert", "list_batches"]
print({k: int(v) for k, v in zip(names, raw[:13])})

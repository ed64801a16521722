"""On the GPU box: lifetime distribution of the 3DGRT forward's workgroups (8x8-pixel ray packets) on the c3_grt_1m_800 frame — how much
of the kernel's duration is a tail of few long-running packets."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["GRUT_GRT_COUNT"] = "1"
import torch
from scenes import torch_batch
syn = importlib.import_module("workloads.synthetic"); grt = importlib.import_module("3dgrut_amd.grt_tracer"); abi = importlib.import_module("3dgrut_amd._abi")
n, W, H = 1_000_000, 800, 800
d12, sph = syn.cloud_trained_like(n, seed=42, median_scale=0.01)
K = syn.pinhole_intrinsics(W, H); ro, rd = syn.pinhole_rays(W, H, K)
batch = torch_batch(dict(rays_ori=ro, rays_dir=rd, T_to_world=syn.orbit_pose(0, n_views=8)[None], intrinsics=K), "cuda")
tr = grt.Tracer({"render": {}}); nat = tr.tracer_wrapper
g = syn.SimpleGaussians(d12, sph)
tr.build_acc(g, rebuild=True)
for _ in range(2):
    out = tr.render(g, batch, train=False)
torch.cuda.synchronize()
nblk = ((W + 63) // 64) * ((H + 63) // 64) * 64 + 512
buf = torch.zeros(16 + 3 * nblk, dtype=torch.int64, device="cuda")
abi.check(nat.lib.grt_debug_fetch_work(nat.handle, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(buf.data_ptr()), buf.numel()), "fetch")
torch.cuda.synchronize()
raw = buf.cpu().numpy()
w = raw[16:].reshape(-1, 3)
ok = w[:, 1] > 0
start, life = w[ok, 0].astype(np.float64) * 0.01, w[ok, 1].astype(np.float64) * 0.01   # us
nodes, leaves = (w[ok, 2] >> 32).astype(np.float64), (w[ok, 2] & 0xFFFFFFFF).astype(np.float64)
t0, t1 = start.min(), (start + life).max()
span = t1 - t0
print(f"packets {ok.sum()}, kernel span {span / 1e3:.2f} ms; packet lifetime mean {life.mean() / 1e3:.2f} p50 {np.percentile(life, 50) / 1e3:.2f} "
      f"p90 {np.percentile(life, 90) / 1e3:.2f} p99 {np.percentile(life, 99) / 1e3:.2f} max {life.max() / 1e3:.2f} ms")
print("sum of lifetimes / (span x 4096 slots) = %.3f" % (life.sum() / (span * 4096)))
edges = np.linspace(t0, t1, 21)
occ = [(np.minimum(start + life, edges[i + 1]) - np.maximum(start, edges[i])).clip(min=0).sum() / ((edges[i + 1] - edges[i]) * 4096) for i in range(20)]
print("slot occupancy over the kernel's duration (20 bins):", " ".join("%.2f" % o for o in occ))
steps = nodes + leaves
print("node + leaf visits per packet: mean %.0f p50 %.0f p99 %.0f max %.0f; correlation with lifetime %.3f; us per visit (least squares through 0) %.3f" % (
    steps.mean(), np.percentile(steps, 50), np.percentile(steps, 99), steps.max(), np.corrcoef(steps, life)[0, 1], (steps * life).sum() / (steps * steps).sum()))
late = np.argsort(start + life)[-40:]
print("the last 40 packets to finish: lifetime mean %.2f ms, start at %.2f of the span, visits mean %.0f" % (life[late].mean() / 1e3, ((start[late] - t0) / span).mean(), steps[late].mean()))
# what a longest-first order would give: greedy list scheduling of the measured lifetimes on 4096 slots
import heapq
for name, order in (("launch order", np.argsort(start)), ("longest first", np.argsort(-life))):
    slots = [0.0] * 4096
    heapq.heapify(slots)
    for i in order:
        t = heapq.heappop(slots)
        heapq.heappush(slots, t + life[i])
    print(f"greedy schedule of the measured lifetimes, {name}: makespan {max(slots) / 1e3:.2f} ms")
names = ["wave_node_visits", "lane_tests", "processed_hits", "lane_rounds", "lane_inserts", "lane_passed_all", "lane_rej_t_range", "lane_rej_box", "lane_rej_distance",
         "wave_tests", "wave_tests_with_a_lane_in_t_range", "wave_tests_with_an_insert", "list_batches"]
print({k: int(v) for k, v in zip(names, raw[:13])})
ph = raw[13:16].astype(np.float64)
print("phase ticks (lane 0 of every packet, 100 MHz): trace rounds %.3f, log merge %.3f, hit processing %.3f of their sum; work-loop (candidate tests) ticks / trace-round ticks = %.3f" % (
    ph[0] / ph.sum(), ph[1] / ph.sum(), ph[2] / ph.sum(), raw[0] / max(ph[0], 1.0)))

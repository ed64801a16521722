"""GPU box: 3DGRT backward at 1 M / 800x800 on a ray subsample against the oracle — default (log replay), exact (tree walk), default without lists."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle
import parity_util as pu
from scenes import rel_err, torch_batch
syn = importlib.import_module("workloads.synthetic"); grt = importlib.import_module("3dgrut_amd.grt_tracer")
n, w, h, ms = 1_000_000, 800, 800, 0.01
stride = int(sys.argv[1]) if len(sys.argv) > 1 else 149
inp = pu.make_frame_inputs(n, w, h, ms)
d12, sph = inp["d12"], inp["sph"]
g = syn.SimpleGaussians(d12, sph)
batch = torch_batch(inp["batch"], "cuda")
sel = np.arange(0, w * h, stride)
rng = np.random.default_rng(4)
g_rad = np.zeros((h * w, 3), np.float32); g_dns = np.zeros((h * w, 1), np.float32)
g_rad[sel] = rng.normal(size=(sel.size, 3)); g_dns[sel] = rng.normal(size=(sel.size, 1))

def hip(env=None, **kw):
    for k, v in (env or {}).items(): os.environ[k] = v
    tr = grt.Tracer({"render": dict(enable_hitcounts=True, **kw)})
    tr.build_acc(g, rebuild=True)
    g.zero_grad()
    out = tr.render(g, batch, train=True)
    loss = (out["pred_features"][0] * torch.as_tensor(g_rad.reshape(h, w, 3), device="cuda")).sum() + (out["pred_opacity"][0] * torch.as_tensor(g_dns.reshape(h, w, 1), device="cuda")).sum()
    loss.backward(); torch.cuda.synchronize()
    st = tr.tracer_wrapper.stats()
    for k in (env or {}): del os.environ[k]
    return g.grads_packed(), st, tr

# the oracle's forward first: compositing flips (rays whose hit number / accepted count differs) carry no upstream gradient
tr0 = grt.Tracer({"render": dict(enable_hitcounts=True)})
tr0.build_acc(g, rebuild=True)
nat0 = tr0.tracer_wrapper
frame = nat0.make_frame(0, 3, tr0._min_transmittance, n, h, w, batch.T_to_world)
res = nat0.trace(frame, torch.as_tensor(d12, device="cuda").contiguous(), torch.as_tensor(sph, device="cuda").contiguous(), batch.rays_ori.contiguous(), batch.rays_dir.contiguous(), hit_capacity=8)
num = res[7].cpu().numpy().astype(np.int64); cnt = res[4][0].cpu().numpy().reshape(-1)
inst = nat0.instances(n, "cuda").cpu().numpy(); scene = np.array(list(nat0.stats().scene_aabb), np.float32)
cfg = oracle.default_grt_config()
ro, rd = inp["rays"]
ora = oracle.grt_forward(cfg, d12, sph, 3, tr0._min_transmittance, inp["batch"]["T_to_world"][0], ro.reshape(-1, 3)[sel][None], rd.reshape(-1, 3)[sel][None], inst=inst, scene=scene, dbg_cap=8)
F = (num[sel] != ora["hit_num"].astype(np.int64)) | (cnt[sel] != ora["hit_count"].reshape(-1))
print("flip rays", int(F.sum()), "of", sel.size)
g_rad[sel[F]] = 0; g_dns[sel[F]] = 0
(gd_def, gs_def), st, tr = hip()
print("default: rederived", st.bwd_rederived_rays, "premise", st.bwd_premise_rays)
(gd_ex, gs_ex), _, _ = hip(backward_hit_replay=False)
(gd_nl, gs_nl), st2, _ = hip(env={"GRUT_GRT_NO_LISTS": "1"})
print("no-lists default: rederived", st2.bwd_rederived_rays, "premise", st2.bwd_premise_rays)
ora["rays"] = (ora["rays"][0].reshape(1, -1, 3), ora["rays"][1].reshape(1, -1, 3))
sh = np.zeros(sel.size, np.uint8)
rdg, rsg = oracle.grt_backward(cfg, 3, tr._min_transmittance, ora, g_rad[sel][None], g_dns[sel][None], np.zeros((1, sel.size, 1), np.float32), round_shift=sh)
print("oracle: rays", sel.size, "round-shifted", int(sh.sum()))
for name, (a, b) in dict(default=(gd_def, gs_def), exact=(gd_ex, gs_ex), nolists=(gd_nl, gs_nl)).items():
    print(name, {k: f"{rel_err(a[:, sl], rdg[:, sl]):.2e}" for k, sl in pu.GRAD_SLICES.items()}, f"sph {rel_err(b, rsg):.2e}")
print("default vs exact", {k: f"{rel_err(gd_def[:, sl], gd_ex[:, sl]):.2e}" for k, sl in pu.GRAD_SLICES.items()})

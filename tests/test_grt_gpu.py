"""GPU parity: the HIP 3DGRT path (LBVH + software traversal, through the C-ABI and the Tracer plugin surface) against
the CPU oracle.  The per-ray ORDER of processed particles is compared bit-exactly: both sides evaluate the candidate test
and the hit distance with the same fp32 expressions on the same proxy instance records (fetched from the GPU build)."""
import importlib

import numpy as np
import pytest

import oracle
from scenes import make_scene, rel_err, torch_batch

pytestmark = pytest.mark.gpu
syn = importlib.import_module("workloads.synthetic")


def _tracer(**render_kw):
    gt = importlib.import_module("3dgrut_amd.grt_tracer")
    return gt.Tracer({"render": render_kw})


def _scene(n, w, h, scale, max_density=0.8, kind="trained"):
    scene = make_scene(n=n, width=w, height=h, median_scale=scale, max_density=max_density, kind=kind)
    scene["T"] = scene["batch"]["T_to_world"][0]
    return scene


def _gpu_hits(scene, cap=256, **kw):
    import torch
    tr = _tracer(**kw)
    g = syn.SimpleGaussians(scene["density12"], scene["sph"])
    tr.build_acc(g, rebuild=True)
    nat = tr.tracer_wrapper
    batch = torch_batch(scene["batch"], "cuda")
    H, W = scene["H"], scene["W"]
    frame = nat.make_frame(0, 3, tr._min_transmittance, g.num_gaussians, H, W, batch.T_to_world)
    d12 = torch.as_tensor(scene["density12"], device="cuda").contiguous()
    sph = torch.as_tensor(scene["sph"], device="cuda").contiguous()
    res = nat.trace(frame, d12, sph, batch.rays_ori.contiguous(), batch.rays_dir.contiguous(), hit_capacity=cap)
    inst = nat.instances(g.num_gaussians, "cuda").cpu().numpy()
    scene_aabb = np.array(list(nat.stats().scene_aabb), np.float32)
    torch.cuda.synchronize()
    return tr, [t.cpu().numpy() for t in res], inst, scene_aabb


@pytest.mark.parametrize("n,w,h,scale,kind", [(500, 40, 24, 0.12, "trained"), (4000, 64, 48, 0.06, "trained"), (1, 16, 16, 0.3, "trained"),
                                              (2, 16, 16, 0.3, "trained"), (3000, 48, 32, 0.05, "random")])
def test_hit_order_is_bit_exact(n, w, h, scale, kind):
    scene = _scene(n, w, h, scale, kind=kind)
    tr, (feat, dns, hit, nrm, cnt, vis, ids, num), inst, scene_aabb = _gpu_hits(scene)
    cfg = oracle.default_grt_config()
    ora = oracle.grt_forward(cfg, scene["density12"], scene["sph"], 3, 1e-3, scene["T"], *scene["rays"], inst=inst, scene=scene_aabb, dbg_cap=256)
    num = num.astype(np.int64)
    assert np.array_equal(num, ora["hit_num"].astype(np.int64)), f"{(num != ora['hit_num']).sum()} rays with a different number of processed hits"
    k = np.minimum(num, 256)
    got, ref = ids.view(np.uint32), ora["hit_ids"]
    for r in range(h * w):
        assert np.array_equal(got[r, :k[r]], ref[r, :k[r]]), f"ray {r}: order differs"
    assert num.max() > 0 or n <= 2
    # compositing of identical hit lists: images agree to rounding
    assert np.abs(feat[0] - ora["features"]).max() < 1e-4 and np.abs(dns[0] - ora["density"]).max() < 1e-4
    assert np.abs(hit[0] - ora["hit_distance"]).max() < 1e-4 * max(1.0, float(np.abs(ora["hit_distance"]).max()))
    assert np.array_equal(cnt[0], ora["hit_count"])
    assert np.array_equal(vis.view(np.int32).reshape(-1) != 0, ora["visibility"] != 0)


def test_proxies_match_oracle():
    scene = _scene(2000, 16, 16, 0.08)
    tr, _, inst, scene_aabb = _gpu_hits(scene)
    d12 = scene["density12"]
    pr = oracle.grt_proxies(oracle.default_grt_config(), d12[:, 0:3], d12[:, 4:8], d12[:, 8:11], d12[:, 3])
    assert rel_err(inst, pr["inst"]) < 2e-6
    ext = pr["scene"][3:] - pr["scene"][:3]
    assert np.all(scene_aabb[:3] <= pr["scene"][:3] + 1e-6) and np.all(scene_aabb[3:] >= pr["scene"][3:] - 1e-6)  # padded, never smaller
    assert np.all(np.abs(scene_aabb - pr["scene"]) < 2e-3 * ext.max())


def _render(scene, g_rad=None, g_dns=None, g_hit=None, **kw):
    import torch
    tr = _tracer(**kw)
    g = syn.SimpleGaussians(scene["density12"], scene["sph"])
    tr.build_acc(g, rebuild=True)
    out = tr.render(g, torch_batch(scene["batch"], "cuda"), train=True)
    res = dict(out=out, tracer=tr)
    if g_rad is not None:
        loss = (out["pred_features"][0] * torch.as_tensor(g_rad, device="cuda")).sum() + (out["pred_opacity"][0] * torch.as_tensor(g_dns, device="cuda")).sum()
        if g_hit is not None:
            loss = loss + (out["pred_dist"][0] * torch.as_tensor(g_hit, device="cuda")).sum()
        loss.backward()
        res["grads"] = g.grads_packed()
    torch.cuda.synchronize()
    return res


@pytest.mark.parametrize("n,w,h,scale,with_depth_grad", [(800, 48, 32, 0.1, True), (800, 48, 32, 0.1, False), (6000, 64, 40, 0.05, False)])
def test_render_and_gradients_match_oracle(n, w, h, scale, with_depth_grad):
    scene = _scene(n, w, h, scale)
    rng = np.random.default_rng(4)
    g_rad = rng.normal(size=(h, w, 3)).astype(np.float32)
    g_dns = rng.normal(size=(h, w, 1)).astype(np.float32)
    g_hit = (rng.normal(size=(h, w, 1)) * 0.1).astype(np.float32)
    gpu = _render(scene, g_rad, g_dns, g_hit if with_depth_grad else None)
    cfg = oracle.default_grt_config()
    ora = oracle.grt_forward(cfg, scene["density12"], scene["sph"], 3, 1e-3, scene["T"], *scene["rays"])
    out = gpu["out"]
    f = out["pred_features"][0].detach().cpu().numpy()
    o = out["pred_opacity"][0].detach().cpu().numpy()
    d = out["pred_dist"][0].detach().cpu().numpy()
    bad = (np.abs(f - ora["features"]).max(-1) > 1e-4) | (np.abs(o - ora["density"])[..., 0] > 1e-4) | \
          (np.abs(d - ora["hit_distance"][..., :1])[..., 0] > 1e-4 * np.maximum(1.0, ora["hit_distance"][..., 0]))
    assert bad.mean() <= 5e-3, f"{bad.sum()} pixels beyond tolerance"
    rd, rs = oracle.grt_backward(cfg, 3, 1e-3, ora, g_rad, g_dns, g_hit if with_depth_grad else np.zeros_like(g_hit))
    f64 = oracle.grt_forward(cfg, scene["density12"], scene["sph"], 3, 1e-3, scene["T"], *scene["rays"], dtype=np.float64)
    rd64, rs64 = oracle.grt_backward(cfg, 3, 1e-3, f64, g_rad, g_dns, g_hit if with_depth_grad else np.zeros_like(g_hit), dtype=np.float64)
    gd, gs = gpu["grads"]
    n_flip = int(bad.sum()) + int((out["hits_count"][0, ..., 0].detach().cpu().numpy() != ora["hit_count"][..., 0]).sum())

    def trimmed(a, b, drop):
        e = np.abs(np.asarray(a, np.float64) - b).reshape(a.shape[0], -1).max(1)
        e = np.sort(e)[: max(1, len(e) - drop)]
        return float(e.max() / (np.abs(b).max() + 1e-12))

    for name, sl in {"position": slice(0, 3), "density": slice(3, 4), "rotation": slice(4, 8), "scale": slice(8, 11)}.items():
        e = min(trimmed(gd[:, sl], rd[:, sl], 3 * n_flip), trimmed(gd[:, sl], rd64[:, sl], 3 * n_flip))
        assert e < 1e-3, f"grad {name}: rel err {e:.3e}"
    assert min(trimmed(gs, rs, 3 * n_flip), trimmed(gs, rs64, 3 * n_flip)) < 1e-3


def test_backward_replay_and_traversal_fallback_agree(monkeypatch):
    """The backward normally replays the forward's hit log; if the log overflows it traverses again.  Both must give the
    same gradients (same hits, same order) up to atomic summation order."""
    scene = _scene(3000, 48, 32, 0.06)
    rng = np.random.default_rng(8)
    g_rad = rng.normal(size=(32, 48, 3)).astype(np.float32)
    g_dns = rng.normal(size=(32, 48, 1)).astype(np.float32)
    a = _render(scene, g_rad, g_dns)["grads"]
    monkeypatch.setenv("GRUT_GRT_LOG_CHUNKS", "3")   # far too small: every frame overflows
    b = _render(scene, g_rad, g_dns)["grads"]
    assert rel_err(a[0], b[0]) < 1e-5 and rel_err(a[1], b[1]) < 1e-5
    assert np.abs(b[0]).max() > 0


@pytest.mark.parametrize("no_lists", [False, True])
def test_default_backward_is_the_reference_backward_program(monkeypatch, no_lists):
    """The default backward replays the forward's hit log except on the rays the forward flags (a processed hit whose proxy box
    the ray enters beyond endT is never offered to the reference's backward trace, which shifts its later rounds of 16,
    referenceBwdOptix.cu:123-131); those get their rounds re-derived — over the frame's packet lists, or by the tree walk when there
    are none.  On a scene with such rays (long, thin, overlapping proxies) the result must be the exact-traversal backward's."""
    if no_lists:
        monkeypatch.setenv("GRUT_GRT_NO_LISTS", "1")
    scene = _scene(2500, 64, 48, 0.09)
    rng = np.random.default_rng(11)
    g_rad = rng.normal(size=(48, 64, 3)).astype(np.float32)
    g_dns = rng.normal(size=(48, 64, 1)).astype(np.float32)
    cfg = oracle.default_grt_config()
    ora = oracle.grt_forward(cfg, scene["density12"], scene["sph"], 3, 1e-3, scene["T"], *scene["rays"])
    ora["rays"] = (ora["rays"][0].reshape(1, -1, 3), ora["rays"][1].reshape(1, -1, 3))
    shifted = np.zeros(48 * 64, np.uint8)
    oracle.grt_backward(cfg, 3, 1e-3, ora, g_rad.reshape(1, -1, 3), g_dns.reshape(1, -1, 1), np.zeros((1, 48 * 64, 1), np.float32), round_shift=shifted)
    assert shifted.sum() >= 3, "the scene must contain round-shifted rays for this test to mean anything"
    default = _render(scene, g_rad, g_dns)["grads"]
    exact = _render(scene, g_rad, g_dns, backward_hit_replay=False)["grads"]
    assert rel_err(default[0], exact[0]) < 2e-5 and rel_err(default[1], exact[1]) < 2e-5   # same hits in the same rounds; atomics order differs
    assert np.abs(exact[0]).max() > 0


def test_refit_update_matches_full_rebuild():
    """rebuild=False keeps the tree topology and refits the boxes (OPTIX_BUILD_OPERATION_UPDATE); results must not change."""
    import torch
    scene = _scene(3000, 48, 32, 0.06)
    tr = _tracer(particle_kernel_density_clamping=False)
    g = syn.SimpleGaussians(scene["density12"], scene["sph"])
    batch = torch_batch(scene["batch"], "cuda")
    tr.build_acc(g, rebuild=True)
    with torch.no_grad():
        g.positions += 0.01 * torch.randn_like(g.positions)
    tr.build_acc(g, rebuild=False)
    a = tr.render(g, batch)["pred_features"].detach().clone()
    tr.build_acc(g, rebuild=True)
    b = tr.render(g, batch)["pred_features"].detach()
    assert torch.allclose(a, b, atol=1e-6)


def test_errors_are_loud():
    gt = importlib.import_module("3dgrut_amd.grt_tracer")
    with pytest.raises(NotImplementedError):
        gt.Tracer({"render": {"primitive_type": "dodecahedron"}})
    tr = _tracer()
    scene = _scene(10, 8, 8, 0.2)
    g = syn.SimpleGaussians(scene["density12"], scene["sph"])
    with pytest.raises(RuntimeError, match="build_bvh"):
        tr.render(g, torch_batch(scene["batch"], "cuda"))


def test_one_tracer_across_different_scenes():
    """Grow-only handle: a small scene, a larger one (BVH buffers and hit log regrow), then the small one again; every frame
    against the oracle."""
    import torch
    tr = _tracer()
    cfg = oracle.default_grt_config()
    for n, w, h, scale, seed in [(300, 24, 16, 0.12, 21), (5000, 72, 48, 0.05, 22), (300, 24, 16, 0.12, 21)]:
        scene = make_scene(n=n, width=w, height=h, median_scale=scale, max_density=0.8, seed=seed)
        g = syn.SimpleGaussians(scene["density12"], scene["sph"])
        tr.build_acc(g, rebuild=True)
        out = tr.render(g, torch_batch(scene["batch"], "cuda"), train=True)
        (out["pred_features"].sum() + out["pred_opacity"].sum()).backward()
        torch.cuda.synchronize()
        ora = oracle.grt_forward(cfg, scene["density12"], scene["sph"], 3, 1e-3, scene["batch"]["T_to_world"][0], *scene["rays"])
        f = out["pred_features"][0].detach().cpu().numpy()
        bad = np.abs(f - ora["features"]).max(-1) > 1e-4
        assert bad.mean() <= 5e-3, f"n={n}: {bad.sum()} pixels beyond tolerance"
        gr = oracle.grt_backward(cfg, 3, 1e-3, ora, np.ones((h, w, 3), np.float32), np.ones((h, w, 1), np.float32), np.zeros((h, w, 1), np.float32))
        gd, gs = g.grads_packed()
        assert rel_err(gd[:, :11], gr[0][:, :11]) < 5e-3 and rel_err(gs, gr[1]) < 5e-3


def test_runs_on_the_callers_stream():
    """BVH build, forward and backward on a side stream: same image, gradients equal up to the order of float atomics."""
    import torch
    scene = _scene(2000, 48, 32, 0.07)
    rng = np.random.default_rng(2)
    g_rad = rng.normal(size=(32, 48, 3)).astype(np.float32)
    g_dns = rng.normal(size=(32, 48, 1)).astype(np.float32)
    ref = _render(scene, g_rad, g_dns)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        got = _render(scene, g_rad, g_dns)
    side.synchronize()
    assert torch.equal(ref["out"]["pred_features"], got["out"]["pred_features"])
    assert rel_err(got["grads"][0], ref["grads"][0]) < 1e-5 and rel_err(got["grads"][1], ref["grads"][1]) < 1e-5


def test_render_matches_reference_programs_golden():
    """The HIP 3DGRT path DIRECTLY against tests/golden/grt_trace.npz — the reference's own forward / backward OptiX programs run
    on the host over an emulated traversal (oracle/ref/ref_grt_trace*.cpp): accepted-hit counts, visibility, images and gradients."""
    import os
    import sys
    import torch
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden
    g = np.load(os.path.join(here, "golden", "grt_trace.npz"))
    for k, kw in enumerate(make_golden.GRT_TRACE_SCENES):
        scene = make_scene(**kw)
        H, W = kw["height"], kw["width"]
        g_rad, g_dns, g_hit = make_golden.grt_trace_upstream(H, W)
        gpu = _render(scene, g_rad, g_dns, g_hit, enable_normals=True)
        out = gpu["out"]
        cnt = out["hits_count"][0].detach().cpu().numpy()
        flips = cnt != g[f"s{k}_hits_count"]
        assert flips.mean() <= 0.01, f"scene {k}: {int(flips.sum())} rays with a different number of accepted hits"
        ok = ~flips[..., 0]
        assert np.abs(out["pred_features"][0].detach().cpu().numpy() - g[f"s{k}_features"])[ok].max() < 1e-4
        assert np.abs(out["pred_opacity"][0].detach().cpu().numpy() - g[f"s{k}_density"])[ok].max() < 1e-4
        hd = g[f"s{k}_hit_distance"]
        assert np.abs(out["pred_dist"][0].detach().cpu().numpy() - hd[..., :1])[ok].max() <= 1e-4 * max(1.0, np.abs(hd).max())
        vis = out["mog_visibility"].view(-1).view(torch.int32).cpu().numpy() != 0
        assert (vis != (g[f"s{k}_visibility"] != 0)).sum() <= 3 * int(flips.sum())
        if not flips.any():
            gd, gs = gpu["grads"]
            rd, rs = g[f"s{k}_grad_density"], g[f"s{k}_grad_sph"]
            assert rel_err(gd[:, :11], rd[:, :11]) < 1e-3 and rel_err(gs, rs) < 1e-3


# ---- triangle-mesh proxies (render.primitive_type icosahedron / octahedron / tetrahedron / diamond) --------------------------
MESH_PRIMS = {"icosahedron": 1, "octahedron": 2, "tetrahedron": 3, "diamond": 4}


@pytest.mark.parametrize("prim", list(MESH_PRIMS))
def test_mesh_proxies_match_reference_programs_golden(prim):
    """render.primitive_type = icosahedron (the reference paper's own configuration, configs/paper/3dgrt/base_ours_reference.yaml:16) and the
    other closed triangle meshes, DIRECTLY against tests/golden/grt_trace_mesh.npz = the reference's forward / backward programs compiled
    for that primitive over the emulated OptiX (triangles from the reference's mesh kernels, back faces culled): accepted-hit counts,
    images, visibility, gradients.  (Rays whose hit sequences hold two entry distances that tie to rounding may process them in the other
    order: same count, bounded in number.)"""
    import os
    import sys
    import torch
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden
    g = np.load(os.path.join(here, "golden", "grt_trace_mesh.npz"))
    for k, kw in enumerate(make_golden.GRT_TRACE_SCENES[:2 if prim == "icosahedron" else 1]):
        scene = make_scene(**kw)
        H, W = kw["height"], kw["width"]
        g_rad, g_dns, g_hit = make_golden.grt_trace_upstream(H, W)
        gpu = _render(scene, g_rad, g_dns, g_hit, primitive_type=prim)
        out = gpu["out"]
        assert int(gpu["tracer"].tracer_wrapper.stats().list_entries) > 0      # (one ray origin: the packet lists serve the mesh proxies too)
        cnt = out["hits_count"][0].detach().cpu().numpy()
        flips = (cnt != g[f"{prim}_s{k}_hits_count"])[..., 0]
        assert flips.mean() <= 0.01, f"scene {k}: {int(flips.sum())} rays with a different number of accepted hits"
        e = np.abs(out["pred_features"][0].detach().cpu().numpy() - g[f"{prim}_s{k}_features"]).max(-1)
        hd = g[f"{prim}_s{k}_hit_distance"]
        e_d = np.abs(out["pred_dist"][0].detach().cpu().numpy() - hd[..., :1])[..., 0]
        tied = ~flips & ((e > 1e-4) | (e_d > 1e-4 * max(1.0, np.abs(hd).max())))
        assert tied.mean() <= 0.02 and (not tied.any() or e[tied].max() < 5e-2)
        ok = ~flips & ~tied
        assert np.abs(out["pred_opacity"][0].detach().cpu().numpy() - g[f"{prim}_s{k}_density"])[ok].max() < 1e-4
        vis = out["mog_visibility"].view(-1).view(torch.int32).cpu().numpy() != 0
        ndrop = 3 * int((flips | tied).sum())
        assert (vis != (g[f"{prim}_s{k}_visibility"] != 0)).sum() <= ndrop
        gd, gs = gpu["grads"]
        rd, rs = g[f"{prim}_s{k}_grad_density"], g[f"{prim}_s{k}_grad_sph"]
        per = np.sort(np.abs(gd[:, :11].astype(np.float64) - rd[:, :11]).max(1))[: max(1, len(gd) - ndrop)]
        assert per.max() / np.abs(rd[:, :11]).max() < 1e-3
        per = np.sort(np.abs(gs.astype(np.float64) - rs).max(1))[: max(1, len(gs) - ndrop)]
        assert per.max() / np.abs(rs).max() < 1e-3


@pytest.mark.parametrize("prim", ["icosahedron", "tetrahedron"])
def test_mesh_proxies_hit_order_equals_oracle(prim):
    """A larger scene through the debug hit lists: the per-ray SEQUENCE of processed particles (entry-distance order, (t, id) ties) against
    the oracle given the GPU-built proxy records - the candidate test is the same arithmetic operation by operation (face-plane clip in the
    proxy's frame, csrc/grt_polyhedra.inl = oracle/orc_polyhedra.h) - then images and gradients."""
    scene = _scene(4000, 64, 48, 0.06)
    tr, (feat, dns, hit, nrm, cnt, vis, ids, num), inst, scene_aabb = _gpu_hits(scene, primitive_type=prim)
    cfg = oracle.default_grt_config(primitive_type=MESH_PRIMS[prim])
    ora = oracle.grt_forward(cfg, scene["density12"], scene["sph"], 3, 1e-3, scene["T"], *scene["rays"], inst=inst, scene=scene_aabb, dbg_cap=256)
    num = num.astype(np.int64)
    assert np.array_equal(num, ora["hit_num"].astype(np.int64)), f"{(num != ora['hit_num']).sum()} rays with a different number of processed hits"
    k = np.minimum(num, 256)
    got, ref = ids.view(np.uint32), ora["hit_ids"]
    for r in range(scene["H"] * scene["W"]):
        assert np.array_equal(got[r, :k[r]], ref[r, :k[r]]), f"ray {r}: order differs"
    assert num.max() > 20
    assert np.abs(feat[0] - ora["features"]).max() < 1e-4 and np.abs(dns[0] - ora["density"]).max() < 1e-4 and np.array_equal(cnt[0], ora["hit_count"])
    rng = np.random.default_rng(4)
    g_rad = rng.normal(size=(scene["H"], scene["W"], 3)).astype(np.float32)
    g_dns = rng.normal(size=(scene["H"], scene["W"], 1)).astype(np.float32)
    gpu = _render(scene, g_rad, g_dns, None, primitive_type=prim)
    rd, rs = oracle.grt_backward(cfg, 3, 1e-3, ora, g_rad, g_dns, np.zeros_like(g_dns))
    gd, gs = gpu["grads"]
    assert rel_err(gd[:, :11], rd[:, :11]) < 1e-3 and rel_err(gs, rs) < 1e-3


@pytest.mark.parametrize("prim", ["icosahedron", "octahedron", "tetrahedron", "diamond", "trisurfel", "trihexa", "custom", "sphere"])
def test_mesh_proxy_packet_lists_equal_the_tree_walk(monkeypatch, prim):
    """The packet lists with the mesh proxies: binning by the box of the polyhedron's vertices, entry-distance intervals from the bounding
    sphere until a packet's first test refines them - every output and every ray's sequence of processed particles must equal the tree
    walk's, bit for bit (large and tiny particles, partial packets at the border).  trihexa (round 6): every rhombus is a list entry of its own,
    binned by ITS box - flat along its plane's axis, sqrt 2 along the other two (proxy_extents).  custom (round 6): binned by the particle's WORLD box,
    whose rays the intersection program runs for; hit distances bounded by the scale frame's 3-sigma sphere."""
    scene = _scene(20000, 100, 60, 0.03)
    (a, n_lists), (b, n_walk) = _hits_with(scene, monkeypatch, False, primitive_type=prim), _hits_with(scene, monkeypatch, True, primitive_type=prim)
    assert n_lists > 0 and n_walk == 0
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert a[-1].max() > 10    # processed hits per ray


def test_custom_primitives_match_reference_programs_golden():
    """render.primitive_type = custom DIRECTLY against tests/golden/grt_trace_mesh.npz `custom_*` = the reference's forward / backward
    programs compiled with PARTICLE_PRIMITIVE_TYPE = MOGTracingCustom over the emulated OptiX's custom-primitive boxes (the reference's AABB
    kernel) - both scenes.  The HIP path: the instances' hit point, offered to the rays that cross the particle's WORLD box, within 3 sigma
    (candidate_abe); tree walk (the packet lists bound the oriented proxy, not its world box)."""
    import os
    import sys
    import torch
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden
    g = np.load(os.path.join(here, "golden", "grt_trace_mesh.npz"))
    for k, kw in enumerate(make_golden.GRT_TRACE_SCENES):
        scene = make_scene(**kw)
        H, W = kw["height"], kw["width"]
        g_rad, g_dns, g_hit = make_golden.grt_trace_upstream(H, W)
        gpu = _render(scene, g_rad, g_dns, g_hit, primitive_type="custom")
        out = gpu["out"]
        assert int(gpu["tracer"].tracer_wrapper.stats().list_entries) > 0      # packet lists binned by the world boxes (round 6)
        cnt = out["hits_count"][0].detach().cpu().numpy()
        flips = (cnt != g[f"custom_s{k}_hits_count"])[..., 0]
        assert flips.mean() <= 0.01, f"scene {k}: {int(flips.sum())} rays with a different number of accepted hits"
        e = np.abs(out["pred_features"][0].detach().cpu().numpy() - g[f"custom_s{k}_features"]).max(-1)
        hd = g[f"custom_s{k}_hit_distance"]
        e_d = np.abs(out["pred_dist"][0].detach().cpu().numpy() - hd[..., :1])[..., 0]
        tied = ~flips & ((e > 1e-4) | (e_d > 1e-4 * max(1.0, np.abs(hd).max())))
        assert tied.mean() <= 0.02 and (not tied.any() or e[tied].max() < 5e-2)
        ok = ~flips & ~tied
        assert np.abs(out["pred_opacity"][0].detach().cpu().numpy() - g[f"custom_s{k}_density"])[ok].max() < 1e-4
        vis = out["mog_visibility"].view(-1).view(torch.int32).cpu().numpy() != 0
        ndrop = 3 * int((flips | tied).sum())
        assert (vis != (g[f"custom_s{k}_visibility"] != 0)).sum() <= ndrop
        gd, gs = gpu["grads"]
        rd, rs = g[f"custom_s{k}_grad_density"], g[f"custom_s{k}_grad_sph"]
        per = np.sort(np.abs(gd[:, :11].astype(np.float64) - rd[:, :11]).max(1))[: max(1, len(gd) - ndrop)]
        assert per.max() / np.abs(rd[:, :11]).max() < 1e-3
        per = np.sort(np.abs(gs.astype(np.float64) - rs).max(1))[: max(1, len(gs) - ndrop)]
        assert per.max() / np.abs(rs).max() < 1e-3


def test_custom_primitives_hit_order_equals_oracle():
    """A larger scene through the debug hit lists: every ray's SEQUENCE of processed particles against the oracle given the GPU-built proxy
    records (world-box slab test and 3-sigma test: the same operations in the same order on both sides), images, default (replayed)
    backward against the oracle's backward program."""
    scene = _scene(4000, 64, 48, 0.06)
    tr, (feat, dns, hit, nrm, cnt, vis, ids, num), inst, scene_aabb = _gpu_hits(scene, primitive_type="custom")
    cfg = oracle.default_grt_config(primitive_type=5)
    box8 = tr.tracer_wrapper.custom_boxes(len(scene["density12"]), "cuda").cpu().numpy()
    assert np.abs(box8 - oracle.grt_custom_boxes(cfg, scene["density12"])).max() <= 2e-6 * np.abs(box8).max()   # (kernelScale: device logf / sqrtf)
    ora = oracle.grt_forward(cfg, scene["density12"], scene["sph"], 3, 1e-3, scene["T"], *scene["rays"], inst=inst, scene=scene_aabb, dbg_cap=256, box8=box8)
    num = num.astype(np.int64)
    assert np.array_equal(num, ora["hit_num"].astype(np.int64)), f"{(num != ora['hit_num']).sum()} rays with a different number of processed hits"
    k = np.minimum(num, 256)
    got, ref = ids.view(np.uint32), ora["hit_ids"]
    for r in range(scene["H"] * scene["W"]):
        assert np.array_equal(got[r, :k[r]], ref[r, :k[r]]), f"ray {r}: order differs"
    assert num.max() > 20
    assert np.abs(feat[0] - ora["features"]).max() < 1e-4 and np.abs(dns[0] - ora["density"]).max() < 1e-4 and np.array_equal(cnt[0], ora["hit_count"])
    # another candidate set than the instances of the same scene (world box instead of the oriented cube, the plain 3-sigma test instead of
    # intersectInstanceParticle's)
    _, (_, _, _, _, _, _, _, num_inst), _, _ = _gpu_hits(scene)
    assert num.sum() != num_inst.astype(np.int64).sum()
    rng = np.random.default_rng(4)
    g_rad = rng.normal(size=(scene["H"], scene["W"], 3)).astype(np.float32)
    g_dns = rng.normal(size=(scene["H"], scene["W"], 1)).astype(np.float32)
    gpu = _render(scene, g_rad, g_dns, None, primitive_type="custom")
    rd, rs = oracle.grt_backward(cfg, 3, 1e-3, ora, g_rad, g_dns, np.zeros_like(g_dns))
    gd, gs = gpu["grads"]
    assert rel_err(gd[:, :11], rd[:, :11]) < 1e-3 and rel_err(gs, rs) < 1e-3


def test_trisurfel_matches_reference_programs_golden():
    """render.primitive_type = trisurfel (round 5) DIRECTLY against tests/golden/grt_trace_mesh.npz: trisurfel_* = the reference's forward /
    backward programs compiled with PARTICLE_PRIMITIVE_TYPE = MOGTracingTriSurfel (the SurfelPrimitive branches of processHit / processHitBwd,
    no face culling) over the emulated OptiX walking the two triangles per particle the reference's trisurfel kernel wrote - with the
    hit-distance gradient flowing (the surfel's own d hitT chain) and through both backward paths (log replay, traversal)."""
    import os
    import sys
    import torch
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden
    g = np.load(os.path.join(here, "golden", "grt_trace_mesh.npz"))
    kw = make_golden.GRT_TRACE_SCENES[0]
    scene = make_scene(**kw)
    H, W = kw["height"], kw["width"]
    g_rad, g_dns, g_hit = make_golden.grt_trace_upstream(H, W)
    inst_img = np.load(os.path.join(here, "golden", "grt_trace.npz"))["s0_features"]
    for replay in (True, False):
        gpu = _render(scene, g_rad, g_dns, g_hit, primitive_type="trisurfel", backward_hit_replay=replay)
        out = gpu["out"]
        assert int(gpu["tracer"].tracer_wrapper.stats().list_entries) > 0      # (one ray origin: the packet lists serve the flat proxies too)
        cnt = out["hits_count"][0].detach().cpu().numpy()
        flips = (cnt != g["trisurfel_s0_hits_count"])[..., 0]
        assert flips.mean() <= 0.02 and cnt.max() >= 20, f"{int(flips.sum())} rays with a different number of accepted hits"
        feat = out["pred_features"][0].detach().cpu().numpy()
        e = np.abs(feat - g["trisurfel_s0_features"]).max(-1)
        hd = g["trisurfel_s0_hit_distance"]
        e_d = np.abs(out["pred_dist"][0].detach().cpu().numpy() - hd[..., :1])[..., 0]
        tied = ~flips & ((e > 1e-4) | (e_d > 1e-4 * max(1.0, np.abs(hd).max())))
        assert tied.mean() <= 0.02 and (not tied.any() or e[tied].max() < 5e-2), f"{int(tied.sum())} rays differ with the same hit count"
        ok = ~flips & ~tied
        assert np.abs(out["pred_opacity"][0].detach().cpu().numpy() - g["trisurfel_s0_density"])[ok].max() < 1e-4
        assert np.abs(feat - inst_img).max() > 1e-2       # not the instances' image: the response is evaluated at the plane crossing
        vis = out["mog_visibility"].view(-1).view(torch.int32).cpu().numpy() != 0
        ndrop = 3 * int((flips | tied).sum())
        assert (vis != (g["trisurfel_s0_visibility"] != 0)).sum() <= ndrop
        gd, gs = gpu["grads"]
        rd, rs = g["trisurfel_s0_grad_density"], g["trisurfel_s0_grad_sph"]
        per = np.sort(np.abs(gd[:, :11].astype(np.float64) - rd[:, :11]).max(1))[: max(1, len(gd) - ndrop)]
        assert per.max() / np.abs(rd[:, :11]).max() < 1e-3, (replay, per.max() / np.abs(rd[:, :11]).max())
        per = np.sort(np.abs(gs.astype(np.float64) - rs).max(1))[: max(1, len(gs) - ndrop)]
        assert per.max() / np.abs(rs).max() < 1e-3, replay


def test_trisurfel_hit_order_equals_oracle_and_gradients_follow():
    """A larger scene through the debug hit lists: every ray's SEQUENCE of processed particles (plane-crossing distance, (t, id) ties) against
    the oracle given the GPU-built proxy records - the candidate test is the checker's arithmetic operation by operation - then images,
    accumulated normals (the surfel's own, gaussianParticles.cuh:398-400) and the gradients of both backward paths."""
    scene = _scene(4000, 64, 48, 0.06)
    tr, (feat, dns, hit, nrm, cnt, vis, ids, num), inst, scene_aabb = _gpu_hits(scene, primitive_type="trisurfel", enable_normals=True)
    cfg = oracle.default_grt_config(primitive_type=6, enable_normals=1)
    ora = oracle.grt_forward(cfg, scene["density12"], scene["sph"], 3, 1e-3, scene["T"], *scene["rays"], inst=inst, scene=scene_aabb, dbg_cap=256)
    num = num.astype(np.int64)
    assert np.array_equal(num, ora["hit_num"].astype(np.int64)), f"{(num != ora['hit_num']).sum()} rays with a different number of processed hits"
    k = np.minimum(num, 256)
    got, ref = ids.view(np.uint32), ora["hit_ids"]
    for r in range(scene["H"] * scene["W"]):
        assert np.array_equal(got[r, :k[r]], ref[r, :k[r]]), f"ray {r}: order differs"
    assert num.max() > 20
    # accepted-hit counts: the surfel's response is evaluated at gro + grd (-gro.z / grd.z) - for rays that graze a surfel's plane the
    # quotient amplifies the last bit of grd.z, and a hit at its alpha threshold may fall the other way (identified: the count differs)
    flips = (cnt[0] != ora["hit_count"])[..., 0]
    m = dict(flips=int(flips.sum()), feat=float(np.abs(feat[0] - ora["features"]).max()), dns=float(np.abs(dns[0] - ora["density"]).max()),
             nrm=float(np.abs(nrm[0] - ora["normals"])[~flips].max()), nrm_abs=float(np.abs(ora["normals"]).max()))
    rng = np.random.default_rng(4)
    g_rad = rng.normal(size=(scene["H"], scene["W"], 3)).astype(np.float32)
    g_dns = rng.normal(size=(scene["H"], scene["W"], 1)).astype(np.float32)
    g_hit = rng.normal(size=(scene["H"], scene["W"], 1)).astype(np.float32)
    for a in (g_rad, g_dns, g_hit):
        a[flips] = 0.0
    rd, rs = oracle.grt_backward(cfg, 3, 1e-3, ora, g_rad, g_dns, g_hit)
    for replay in (True, False):
        gpu = _render(scene, g_rad, g_dns, g_hit, primitive_type="trisurfel", backward_hit_replay=replay)
        gd, gs = gpu["grads"]
        m[f"grad_replay_{replay}"] = (rel_err(gd[:, :11], rd[:, :11]), rel_err(gs, rs))
    assert m["flips"] <= max(2, 2e-3 * flips.size) and m["feat"] < 1e-4 and m["dns"] < 1e-4, m
    assert m["nrm"] < 1e-4 and m["nrm_abs"] > 0.05, m
    assert all(max(m[f"grad_replay_{r}"]) < 1e-3 for r in (True, False)), m


def test_trihexa_matches_reference_programs_golden_and_the_oracle():
    """render.primitive_type = trihexa (round 5): three rhombi per particle, back faces culled, a ray is offered the SAME particle up to
    three times.  Every rhombus is a proxy of its own here (3 N leaves).  (i) DIRECTLY against tests/golden/grt_trace_mesh.npz: trihexa_* =
    the reference's forward / backward programs compiled for MOGTracingTriHexa over the emulated OptiX walking the six triangles per
    particle of the reference's trihexa kernel, both backward paths; (ii) every ray's SEQUENCE of processed particles - repeats included -
    against the oracle given the GPU-built proxy records, then images and the gradients of both backward paths."""
    import os
    import sys
    import torch
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden
    g = np.load(os.path.join(here, "golden", "grt_trace_mesh.npz"))
    kw = make_golden.GRT_TRACE_SCENES[0]
    scene = make_scene(**kw)
    H, W = kw["height"], kw["width"]
    g_rad, g_dns, g_hit = make_golden.grt_trace_upstream(H, W)
    for replay in (True, False):
        gpu = _render(scene, g_rad, g_dns, g_hit, primitive_type="trihexa", backward_hit_replay=replay)
        out = gpu["out"]
        assert int(gpu["tracer"].tracer_wrapper.stats().list_entries) > 0      # packet lists over the 3 N rhombi (round 6)
        cnt = out["hits_count"][0].detach().cpu().numpy()
        flips = (cnt != g["trihexa_s0_hits_count"])[..., 0]
        assert flips.mean() <= 0.02 and cnt.max() >= 20, f"{int(flips.sum())} rays with a different number of accepted hits"
        e = np.abs(out["pred_features"][0].detach().cpu().numpy() - g["trihexa_s0_features"]).max(-1)
        hd = g["trihexa_s0_hit_distance"]
        e_d = np.abs(out["pred_dist"][0].detach().cpu().numpy() - hd[..., :1])[..., 0]
        tied = ~flips & ((e > 1e-4) | (e_d > 1e-4 * max(1.0, np.abs(hd).max())))
        assert tied.mean() <= 0.02 and (not tied.any() or e[tied].max() < 5e-2), f"{int(tied.sum())} rays differ with the same hit count"
        ok = ~flips & ~tied
        assert np.abs(out["pred_opacity"][0].detach().cpu().numpy() - g["trihexa_s0_density"])[ok].max() < 1e-4
        vis = out["mog_visibility"].view(-1).view(torch.int32).cpu().numpy() != 0
        ndrop = 3 * int((flips | tied).sum())
        assert (vis != (g["trihexa_s0_visibility"] != 0)).sum() <= ndrop
        gd, gs = gpu["grads"]
        rd, rs = g["trihexa_s0_grad_density"], g["trihexa_s0_grad_sph"]
        per = np.sort(np.abs(gd[:, :11].astype(np.float64) - rd[:, :11]).max(1))[: max(1, len(gd) - ndrop)]
        assert per.max() / np.abs(rd[:, :11]).max() < 1e-3, (replay, per.max() / np.abs(rd[:, :11]).max())
        per = np.sort(np.abs(gs.astype(np.float64) - rs).max(1))[: max(1, len(gs) - ndrop)]
        assert per.max() / np.abs(rs).max() < 1e-3, replay
    # (ii) the sequences
    scene = _scene(4000, 64, 48, 0.06)
    tr, (feat, dns, hit, nrm, cnt, vis, ids, num), inst, scene_aabb = _gpu_hits(scene, primitive_type="trihexa")
    assert inst.shape == (4000, 12)                                  # (one record per particle, not per rhombus)
    cfg = oracle.default_grt_config(primitive_type=7)
    ora = oracle.grt_forward(cfg, scene["density12"], scene["sph"], 3, 1e-3, scene["T"], *scene["rays"], inst=inst, scene=scene_aabb, dbg_cap=256)
    num = num.astype(np.int64)
    assert np.array_equal(num, ora["hit_num"].astype(np.int64)), f"{(num != ora['hit_num']).sum()} rays with a different number of processed hits"
    k = np.minimum(num, 256)
    got, ref = ids.view(np.uint32), ora["hit_ids"]
    repeats = 0
    for r in range(scene["H"] * scene["W"]):
        assert np.array_equal(got[r, :k[r]], ref[r, :k[r]]), f"ray {r}: order differs"
        repeats += int(k[r] - len(np.unique(got[r, :k[r]])))
    assert num.max() > 20 and repeats > 0, "no ray was offered a particle twice"
    flips = (cnt[0] != ora["hit_count"])[..., 0]
    m = dict(flips=int(flips.sum()), feat=float(np.abs(feat[0] - ora["features"]).max()), dns=float(np.abs(dns[0] - ora["density"]).max()))
    rng = np.random.default_rng(4)
    g_rad = rng.normal(size=(scene["H"], scene["W"], 3)).astype(np.float32)
    g_dns = rng.normal(size=(scene["H"], scene["W"], 1)).astype(np.float32)
    for a in (g_rad, g_dns):
        a[flips] = 0.0
    rd, rs = oracle.grt_backward(cfg, 3, 1e-3, ora, g_rad, g_dns, np.zeros_like(g_dns))
    for replay in (True, False):
        gpu = _render(scene, g_rad, g_dns, None, primitive_type="trihexa", backward_hit_replay=replay)
        gd, gs = gpu["grads"]
        m[f"grad_replay_{replay}"] = (rel_err(gd[:, :11], rd[:, :11]), rel_err(gs, rs))
    assert m["flips"] <= max(2, 2e-3 * flips.size) and m["feat"] < 1e-4 and m["dns"] < 1e-4, m
    assert all(max(m[f"grad_replay_{r}"]) < 1e-3 for r in (True, False)), m


def test_sphere_matches_reference_programs_golden_and_the_oracle():
    """render.primitive_type = sphere (round 6; optixTracer.cpp:189-190, 765-781): one OptiX built-in sphere per particle - the any-hit program is
    offered a ray's entry into the sphere and, ignoring it, its exit: every root is a proxy of its own here (2 N leaves).  (i) DIRECTLY against
    tests/golden/grt_trace_sphere.npz = the reference's forward / backward programs compiled for MOGTracingSphere over the emulated OptiX's sphere
    primitive, both scenes, both backward paths; (ii) every ray's SEQUENCE of processed particles - repeats included - against the oracle given
    the GPU-built proxy records, then images and gradients."""
    import os
    import sys
    import torch
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden
    g = np.load(os.path.join(here, "golden", "grt_trace_sphere.npz"))
    for k, kw in enumerate(make_golden.GRT_TRACE_SCENES):
        scene = make_scene(**kw)
        H, W = kw["height"], kw["width"]
        g_rad, g_dns, g_hit = make_golden.grt_trace_upstream(H, W)
        for replay in (True, False):
            gpu = _render(scene, g_rad, g_dns, g_hit, primitive_type="sphere", backward_hit_replay=replay)
            out = gpu["out"]
            assert int(gpu["tracer"].tracer_wrapper.stats().list_entries) > 0      # packet lists over the 2 N roots
            cnt = out["hits_count"][0].detach().cpu().numpy()
            flips = (cnt != g[f"sphere_s{k}_hits_count"])[..., 0]
            assert flips.mean() <= 0.01 and cnt.max() >= 20, f"scene {k}: {int(flips.sum())} rays with a different number of accepted hits"
            e = np.abs(out["pred_features"][0].detach().cpu().numpy() - g[f"sphere_s{k}_features"]).max(-1)
            hd = g[f"sphere_s{k}_hit_distance"]
            e_d = np.abs(out["pred_dist"][0].detach().cpu().numpy() - hd[..., :1])[..., 0]
            tied = ~flips & ((e > 1e-4) | (e_d > 1e-4 * max(1.0, np.abs(hd).max())))
            assert tied.mean() <= 0.01 and (not tied.any() or e[tied].max() < 5e-2), f"scene {k}: {int(tied.sum())} rays differ with the same hit count"
            ok = ~flips & ~tied
            assert np.abs(out["pred_opacity"][0].detach().cpu().numpy() - g[f"sphere_s{k}_density"])[ok].max() < 1e-4
            vis = out["mog_visibility"].view(-1).view(torch.int32).cpu().numpy() != 0
            ndrop = 3 * int((flips | tied).sum())
            assert (vis != (g[f"sphere_s{k}_visibility"] != 0)).sum() <= ndrop
            gd, gs = gpu["grads"]
            rd, rs = g[f"sphere_s{k}_grad_density"], g[f"sphere_s{k}_grad_sph"]
            per = np.sort(np.abs(gd[:, :11].astype(np.float64) - rd[:, :11]).max(1))[: max(1, len(gd) - ndrop)]
            assert per.max() / np.abs(rd[:, :11]).max() < 1e-3, (k, replay, per.max() / np.abs(rd[:, :11]).max())
            per = np.sort(np.abs(gs.astype(np.float64) - rs).max(1))[: max(1, len(gs) - ndrop)]
            assert per.max() / np.abs(rs).max() < 1e-3, (k, replay)
    # (ii) the sequences
    scene = _scene(4000, 64, 48, 0.06)
    tr, (feat, dns, hit, nrm, cnt, vis, ids, num), inst, scene_aabb = _gpu_hits(scene, cap=512, primitive_type="sphere")
    assert inst.shape == (4000, 12)                                  # (one record per particle, not per root)
    cfg = oracle.default_grt_config(primitive_type=8)
    ora = oracle.grt_forward(cfg, scene["density12"], scene["sph"], 3, 1e-3, scene["T"], *scene["rays"], inst=inst, scene=scene_aabb, dbg_cap=512)
    num = num.astype(np.int64)
    assert np.array_equal(num, ora["hit_num"].astype(np.int64)), f"{(num != ora['hit_num']).sum()} rays with a different number of processed hits"
    kk = np.minimum(num, 512)
    got, ref = ids.view(np.uint32), ora["hit_ids"]
    repeats = 0
    for r in range(scene["H"] * scene["W"]):
        assert np.array_equal(got[r, :kk[r]], ref[r, :kk[r]]), f"ray {r}: order differs"
        repeats += int(kk[r] - len(np.unique(got[r, :kk[r]])))
    assert num.max() > 20 and repeats > 0, "no ray was offered a particle twice"
    flips = (cnt[0] != ora["hit_count"])[..., 0]
    m = dict(flips=int(flips.sum()), feat=float(np.abs(feat[0] - ora["features"]).max()), dns=float(np.abs(dns[0] - ora["density"]).max()))
    rng = np.random.default_rng(4)
    g_rad = rng.normal(size=(scene["H"], scene["W"], 3)).astype(np.float32)
    g_dns = rng.normal(size=(scene["H"], scene["W"], 1)).astype(np.float32)
    for a in (g_rad, g_dns):
        a[flips] = 0.0
    rd, rs = oracle.grt_backward(cfg, 3, 1e-3, ora, g_rad, g_dns, np.zeros_like(g_dns))
    for replay in (True, False):
        gpu = _render(scene, g_rad, g_dns, None, primitive_type="sphere", backward_hit_replay=replay)
        gd, gs = gpu["grads"]
        m[f"grad_replay_{replay}"] = (rel_err(gd[:, :11], rd[:, :11]), rel_err(gs, rs))
    print("sphere:", m, "repeats", repeats)
    assert m["flips"] <= max(2, 2e-3 * flips.size) and m["feat"] < 1e-4 and m["dns"] < 1e-4, m
    assert all(max(m[f"grad_replay_{r}"]) < 1e-3 for r in (True, False)), m


def test_barycentric_surfels_forward_matches_reference_program_golden_and_the_oracle():
    """render.pipeline_type = barycentricSurfels with render.primitive_type = trisurfel (round 6): the surfel FORWARD pipeline - ten hits per
    trace, the response from the crossing's squared distance in the proxy frame, depth from the hit distances, the surfel's normals.  (i) DIRECTLY
    against tests/golden/grt_trace_bary.npz = barycentricSurfelsOptix.cu over the emulated OptiX's triangles, both scenes; (ii) every ray's sequence
    of processed particles and the images against the oracle given the GPU-built proxy records, packet lists = tree walk; (iii) the backward is
    refused (the reference ships no backward program for this pipeline)."""
    import os
    import sys
    import torch
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden
    g = np.load(os.path.join(here, "golden", "grt_trace_bary.npz"))
    kw_r = dict(primitive_type="trisurfel", pipeline_type="barycentricSurfels", enable_normals=True)
    for k, kw in enumerate(make_golden.GRT_TRACE_SCENES):
        scene = make_scene(**kw)
        tr = _tracer(**kw_r)
        gs = syn.SimpleGaussians(scene["density12"], scene["sph"], requires_grad=False)
        tr.build_acc(gs, rebuild=True)
        with torch.no_grad():
            out = tr.render(gs, torch_batch(scene["batch"], "cuda"))
        assert int(tr.tracer_wrapper.stats().list_entries) > 0
        cnt = out["hits_count"][0].cpu().numpy()
        flips = (cnt != g[f"bary_s{k}_hits_count"])[..., 0]
        assert flips.mean() <= 0.01 and cnt.max() >= 20, f"scene {k}: {int(flips.sum())} rays with a different number of accepted hits"
        ok = ~flips
        assert np.abs(out["pred_features"][0].cpu().numpy() - g[f"bary_s{k}_features"])[ok].max() < 1e-4
        assert np.abs(out["pred_opacity"][0].cpu().numpy() - g[f"bary_s{k}_density"])[ok].max() < 1e-4
        hd = g[f"bary_s{k}_hit_distance"]
        assert np.abs(out["pred_dist"][0].cpu().numpy() - hd[..., :1])[ok].max() <= 1e-4 * max(1.0, np.abs(hd).max())
        gn = g[f"bary_s{k}_normals"]
        gl = np.linalg.norm(gn, axis=-1, keepdims=True)     # (the plugin returns the integrated normal NORMALISED, tracer.py:346)
        sel = ok & (gl[..., 0] > 0.05)
        assert sel.sum() > 0.5 * ok.sum() and np.abs(out["pred_normals"][0].cpu().numpy() - gn / np.maximum(gl, 1e-12))[sel].max() < 2e-3
        vis = out["mog_visibility"].view(-1).view(torch.int32).cpu().numpy() != 0
        assert (vis != (g[f"bary_s{k}_visibility"] != 0)).sum() <= 3 * int(flips.sum())
    # (ii) sequences against the oracle, lists against the walk
    scene = _scene(4000, 64, 48, 0.06)
    tr, (feat, dns, hit, nrm, cnt, vis, ids, num), inst, scene_aabb = _gpu_hits(scene, **kw_r)
    cfg = oracle.default_grt_config(primitive_type=6, pipeline_type=1, enable_normals=1)
    ora = oracle.grt_forward(cfg, scene["density12"], scene["sph"], 3, 1e-3, scene["T"], *scene["rays"], inst=inst, scene=scene_aabb, dbg_cap=256)
    num = num.astype(np.int64)
    assert np.array_equal(num, ora["hit_num"].astype(np.int64)), f"{(num != ora['hit_num']).sum()} rays with a different number of processed hits"
    kk = np.minimum(num, 256)
    got, ref = ids.view(np.uint32), ora["hit_ids"]
    for r in range(scene["H"] * scene["W"]):
        assert np.array_equal(got[r, :kk[r]], ref[r, :kk[r]]), f"ray {r}: order differs"
    flips = (cnt[0] != ora["hit_count"])[..., 0]
    assert flips.sum() <= max(2, 2e-3 * flips.size) and num.max() > 20
    assert np.abs(feat[0] - ora["features"])[~flips].max() < 1e-4 and np.abs(dns[0] - ora["density"])[~flips].max() < 1e-4
    assert np.abs(nrm[0] - ora["normals"])[~flips].max() < 1e-4
    assert (np.abs(hit[0] - ora["hit_distance"]) / np.maximum(1.0, np.abs(ora["hit_distance"])))[~flips].max() < 1e-4


def test_barycentric_surfels_lists_equal_the_tree_walk_and_the_backward_is_refused(monkeypatch):
    import torch
    kw_r = dict(primitive_type="trisurfel", pipeline_type="barycentricSurfels")
    scene = _scene(20000, 100, 60, 0.03)
    (a, n_lists), (b, n_walk) = _hits_with(scene, monkeypatch, False, **kw_r), _hits_with(scene, monkeypatch, True, **kw_r)
    assert n_lists > 0 and n_walk == 0
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    tr = _tracer(**kw_r)
    gs = syn.SimpleGaussians(scene["density12"], scene["sph"])
    tr.build_acc(gs, rebuild=True)
    out = tr.render(gs, torch_batch(scene["batch"], "cuda"), train=True)
    with pytest.raises(RuntimeError, match="forward only"):
        out["pred_features"].sum().backward()
    grt = importlib.import_module("3dgrut_amd.grt_tracer")
    with pytest.raises(NotImplementedError, match="trisurfel"):
        grt.Tracer({"render": {"pipeline_type": "barycentricSurfels"}})


def test_unsupported_primitives_are_refused():
    grt = importlib.import_module("3dgrut_amd.grt_tracer")
    for prim in ("dodecahedron", "cube"):
        with pytest.raises(NotImplementedError, match="primitive_type"):
            grt.Tracer({"render": {"primitive_type": prim}})


def test_c_abi_refuses_feature_kernels_on_open_proxies():
    """Round 6 (advisor): the plugin refuses neural harmonic features on custom / trisurfel (grt_config_from_conf); the C-ABI must too - trisurfel
    would blend at the volumetric intersection instead of the surfel's plane crossing, the Slang pipeline's custom-primitive test is another
    one than GRUT_PRIM_CUSTOM's.  (trihexa and sphere are served since the feature kernels map the log's proxy ids to particles.)"""
    import ctypes as C
    grt = importlib.import_module("3dgrut_amd.grt_tracer")
    abi = importlib.import_module("3dgrut_amd._abi")
    lib = abi.load_library()
    for prim in ("trisurfel",):
        with pytest.raises(NotImplementedError, match="neural harmonic"):
            grt.grt_config_from_conf({"render": {"pipeline_type": "referenceSlang", "primitive_type": prim}, "model": NHT_CONF})
        cfg = grt.grt_config_from_conf({"render": {"pipeline_type": "referenceSlang"}, "model": NHT_CONF})
        cfg.primitive_type = abi.GRT_PRIMITIVES[prim]
        handle = C.c_void_p()
        status = lib.grt_create(C.byref(cfg), C.byref(handle))
        assert status != 0 and not handle.value, (prim, status)
        assert b"neural harmonic features" in lib.grut_last_error()
    cfg = grt.grt_config_from_conf({"render": {"pipeline_type": "referenceSlang", "primitive_type": "icosahedron"}, "model": NHT_CONF})
    handle = C.c_void_p()
    assert lib.grt_create(C.byref(cfg), C.byref(handle)) == 0
    lib.grt_destroy(handle)


# ---- packet lists (frames with one ray origin) against the tree walk -------------------------------------------------------
def _hits_with(scene, monkeypatch, no_lists, rays_ori=None, rays_dir=None, **render_kw):
    import torch
    if no_lists:
        monkeypatch.setenv("GRUT_GRT_NO_LISTS", "1")
    else:
        monkeypatch.delenv("GRUT_GRT_NO_LISTS", raising=False)
    tr = _tracer(**render_kw)
    g = syn.SimpleGaussians(scene["density12"], scene["sph"])
    tr.build_acc(g, rebuild=True)
    nat = tr.tracer_wrapper
    batch = torch_batch(scene["batch"], "cuda")
    ro = batch.rays_ori.contiguous() if rays_ori is None else torch.as_tensor(rays_ori, device="cuda").contiguous()
    frame = nat.make_frame(0, 3, tr._min_transmittance, g.num_gaussians, scene["H"], scene["W"], batch.T_to_world)
    d12 = torch.as_tensor(scene["density12"], device="cuda").contiguous()
    sph = torch.as_tensor(scene["sph"], device="cuda").contiguous()
    rd = batch.rays_dir.contiguous() if rays_dir is None else torch.as_tensor(rays_dir, device="cuda").contiguous()
    res = nat.trace(frame, d12, sph, ro, rd, hit_capacity=128)
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in res], int(nat.stats().list_entries)


@pytest.mark.parametrize("n,w,h,scale,kind", [(6000, 72, 56, 0.05, "trained"), (3000, 50, 30, 0.08, "random"), (1, 16, 16, 0.3, "trained"),
                                              (40000, 96, 64, 0.02, "trained")])
def test_packet_lists_equal_the_tree_walk(monkeypatch, n, w, h, scale, kind):
    """One ray origin for the whole frame: the forward bins the particles per 8x8 ray packet and scans sorted lists instead of walking the
    BVH.  Same candidate arithmetic, same per-ray buffers: every output and every ray's sequence of processed particles is identical,
    bit for bit (image widths that are no multiple of 8 leave partial packets at the border)."""
    scene = _scene(n, w, h, scale, kind=kind)
    lists, n_entries = _hits_with(scene, monkeypatch, no_lists=False)
    walk, n_walk = _hits_with(scene, monkeypatch, no_lists=True)
    assert n_entries > 0 and n_walk == 0
    for a, b, name in zip(lists, walk, ("features", "density", "hit_distance", "normals", "hit_count", "visibility", "ids", "num")):
        if name == "ids":
            num = lists[7].reshape(-1).astype(np.int64)
            for r in range(h * w):
                k = min(int(num[r]), 128)
                assert np.array_equal(a.reshape(h * w, -1)[r, :k], b.reshape(h * w, -1)[r, :k]), f"ray {r}: order differs"
        else:
            assert np.array_equal(a, b), name


def test_packet_lists_with_the_camera_inside_the_cloud(monkeypatch):
    """Ray origin in the middle of the particles: proxies that contain the apex of the cones (key 0, every packet reaches them) and
    particles behind the camera.  Lists and tree walk still agree bit for bit."""
    scene = _scene(8000, 64, 48, 0.06)
    T = scene["batch"]["T_to_world"].copy()
    T[0, :3, 3] = np.float32(0.05)           # the cloud is centred on the origin
    scene["batch"] = dict(scene["batch"], T_to_world=T)
    scene["T"] = T[0]
    lists, n_entries = _hits_with(scene, monkeypatch, no_lists=False)
    walk, n_walk = _hits_with(scene, monkeypatch, no_lists=True)
    assert n_entries > 0 and n_walk == 0
    num = lists[7].reshape(-1).astype(np.int64)
    assert num.max() > 16                     # several trace rounds per ray
    for a, b, name in zip(lists, walk, ("features", "density", "hit_distance", "normals", "hit_count", "visibility", "ids", "num")):
        if name == "ids":
            for r in range(num.size):
                k = min(int(num[r]), 128)
                assert np.array_equal(a.reshape(num.size, -1)[r, :k], b.reshape(num.size, -1)[r, :k]), f"ray {r}: order differs"
        else:
            assert np.array_equal(a, b), name


def test_packet_lists_with_unnormalised_directions(monkeypatch):
    """Hit distances are ray parameters: with directions of different lengths (0.4 .. 2.5 here, varying per pixel) the lists' distance
    bounds must scale with the frame's smallest / largest |d|.  Lists and tree walk agree bit for bit."""
    scene = _scene(6000, 56, 40, 0.06)
    rng = np.random.default_rng(3)
    rd = (scene["rays"][1] * rng.uniform(0.4, 2.5, size=scene["rays"][1].shape[:-1] + (1,))).astype(np.float32)
    lists, n_entries = _hits_with(scene, monkeypatch, no_lists=False, rays_dir=rd)
    walk, n_walk = _hits_with(scene, monkeypatch, no_lists=True, rays_dir=rd)
    assert n_entries > 0 and n_walk == 0
    num = lists[7].reshape(-1).astype(np.int64)
    assert num.max() > 16
    for a, b, name in zip(lists, walk, ("features", "density", "hit_distance", "normals", "hit_count", "visibility", "ids", "num")):
        if name == "ids":
            for r in range(num.size):
                k = min(int(num[r]), 128)
                assert np.array_equal(a.reshape(num.size, -1)[r, :k], b.reshape(num.size, -1)[r, :k]), f"ray {r}: order differs"
        else:
            assert np.array_equal(a, b), name


def _assert_same_hits(lists, walk):
    num = lists[7].reshape(-1).astype(np.int64)
    for a, b, name in zip(lists, walk, ("features", "density", "hit_distance", "normals", "hit_count", "visibility", "ids", "num")):
        if name == "ids":
            for r in range(num.size):
                k = min(int(num[r]), 128)
                assert np.array_equal(a.reshape(num.size, -1)[r, :k], b.reshape(num.size, -1)[r, :k]), f"ray {r}: order differs"
        else:
            assert np.array_equal(a, b), name


def _fan_rays(h, w, half_angle_deg, roll_deg=0.0):
    """Directions on an equidistant fan (a fisheye): pixel offset from the centre = angle off the axis, rolled about the axis."""
    ys, xs = np.meshgrid(np.arange(h) - (h - 1) / 2, np.arange(w) - (w - 1) / 2, indexing="ij")
    r = np.hypot(xs, ys) + 1e-9
    theta = r / r.max() * np.deg2rad(half_angle_deg)
    phi = np.arctan2(ys, xs) + np.deg2rad(roll_deg)
    d = np.stack([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)], -1)
    return d[None].astype(np.float32)


@pytest.mark.parametrize("half_angle,roll,expect_plane", [(60.0, 33.0, True), (84.0, 0.0, True), (120.0, 10.0, False)])
def test_packet_lists_over_a_fisheye_fan(monkeypatch, half_angle, roll, expect_plane):
    """The binning selects a particle's candidate packets on the frame's tangent plane (GrtGrid) when every ray is within 87 degrees of
    the centre pixel's direction — rolled and strongly distorted grids included — and by scanning the super tiles' cones otherwise
    (a fan of 240 degrees here).  The camera sits inside the cloud, so some proxy boxes straddle the apex plane.  Either way the lists
    hold what the tree walk finds."""
    scene = _scene(8000, 72, 56, 0.06)
    T = scene["batch"]["T_to_world"].copy()
    T[0, :3, 3] = np.float32(0.03)
    scene["batch"] = dict(scene["batch"], T_to_world=T)
    scene["T"] = T[0]
    rd = _fan_rays(56, 72, half_angle, roll)
    lists, n_entries = _hits_with(scene, monkeypatch, no_lists=False, rays_dir=rd)
    walk, n_walk = _hits_with(scene, monkeypatch, no_lists=True, rays_dir=rd)
    assert n_entries > 0 and n_walk == 0
    _assert_same_hits(lists, walk)
    # the super tile scan on the same frame: sound too, and never shorter than the tangent-plane lists (which also drop the packets
    # a separating plane of the FRAME's axes excludes)
    monkeypatch.setenv("GRUT_GRT_NO_GRID", "1")
    scan, n_scan = _hits_with(scene, monkeypatch, no_lists=False, rays_dir=rd)
    _assert_same_hits(scan, walk)
    assert (n_entries <= n_scan) if expect_plane else (n_entries == n_scan)


def test_packet_lists_with_large_and_small_particles(monkeypatch):
    """Particles from a fraction of a packet to a third of the frame in one cloud: the lane-tested rectangles (<= 32 packets), the
    wave-tested ones and the border packets of an image whose sides are no multiples of 8."""
    scene = _scene(5000, 150, 91, 0.03)
    d12 = scene["density12"].copy()
    rng = np.random.default_rng(11)
    big = rng.choice(d12.shape[0], 150, replace=False)
    d12[big, 8:11] *= rng.uniform(4.0, 30.0, size=(150, 1)).astype(np.float32)   # scales
    scene["density12"] = d12
    lists, n_entries = _hits_with(scene, monkeypatch, no_lists=False)
    walk, n_walk = _hits_with(scene, monkeypatch, no_lists=True)
    assert n_entries > 0 and n_walk == 0
    _assert_same_hits(lists, walk)


def test_packet_lists_over_frames_of_changing_size(monkeypatch):
    """One tracer, frames of different sizes, with the speculative list tail on (GRUT_GRT_SPECULATE: expansion, entry sort and ranges are
    enqueued against the capacity of the per-entry buffers while the entry count travels to the host; off by default — it measured no
    faster): a frame that needs more than the buffers hold runs the tail again after growing them, a smaller one leaves the excess
    untouched.  Every frame matches its tree walk."""
    import torch
    monkeypatch.delenv("GRUT_GRT_NO_LISTS", raising=False)
    monkeypatch.setenv("GRUT_GRT_SPECULATE", "1")
    tr = _tracer()
    nat = tr.tracer_wrapper
    for n, w, h, scale in [(3000, 48, 32, 0.05), (20000, 96, 64, 0.04), (1500, 40, 24, 0.08), (20000, 96, 64, 0.04)]:
        scene = _scene(n, w, h, scale)
        g = syn.SimpleGaussians(scene["density12"], scene["sph"])
        tr.build_acc(g, rebuild=True)
        batch = torch_batch(scene["batch"], "cuda")
        frame = nat.make_frame(0, 3, tr._min_transmittance, g.num_gaussians, h, w, batch.T_to_world)
        d12 = torch.as_tensor(scene["density12"], device="cuda").contiguous()
        sph = torch.as_tensor(scene["sph"], device="cuda").contiguous()
        res = nat.trace(frame, d12, sph, batch.rays_ori.contiguous(), batch.rays_dir.contiguous(), hit_capacity=128)
        torch.cuda.synchronize()
        lists = [t.cpu().numpy() for t in res]
        assert int(nat.stats().list_entries) > 0
        walk, n_walk = _hits_with(scene, monkeypatch, no_lists=True)
        monkeypatch.delenv("GRUT_GRT_NO_LISTS", raising=False)
        assert n_walk == 0
        _assert_same_hits(lists, walk)


def test_a_degenerate_direction_sends_the_frame_to_the_tree_walk(monkeypatch):
    """A ray with a zero direction has no place in a bounding cone: the frame is served by the tree walk (decided on the device) and the
    other rays are rendered as if nothing had happened."""
    scene = _scene(3000, 40, 24, 0.08)
    rd = scene["rays"][1].copy()
    rd[0, 3, 5] = 0.0
    res, n_entries = _hits_with(scene, monkeypatch, no_lists=False, rays_dir=rd)
    ref, _ = _hits_with(scene, monkeypatch, no_lists=True, rays_dir=rd)
    assert n_entries == 0
    keep = np.ones((24, 40), bool)
    keep[3, 5] = False
    for a, b, name in zip(res[:5], ref[:5], ("features", "density", "hit_distance", "normals", "hit_count")):
        assert np.array_equal(a[0][keep], b[0][keep]), name
    assert np.isfinite(res[0][0][keep]).all()


def test_rays_with_different_origins_take_the_tree_walk(monkeypatch):
    """Packet lists need one ray origin (the cones have one apex).  A frame in which a single ray starts elsewhere is served by the tree
    walk — decided on the device, reported by grt_stats — and still matches the oracle's hit order."""
    scene = _scene(3000, 40, 24, 0.08)
    ro = scene["rays"][0].copy()
    ro[0, 5, 7] += np.float32(0.01)
    res, n_entries = _hits_with(scene, monkeypatch, no_lists=False, rays_ori=ro)
    assert n_entries == 0
    tr = _tracer()
    g = syn.SimpleGaussians(scene["density12"], scene["sph"])
    tr.build_acc(g, rebuild=True)
    inst = tr.tracer_wrapper.instances(g.num_gaussians, "cuda").cpu().numpy()
    scene_aabb = np.array(list(tr.tracer_wrapper.stats().scene_aabb), np.float32)
    ora = oracle.grt_forward(oracle.default_grt_config(), scene["density12"], scene["sph"], 3, 1e-3, scene["T"], ro, scene["rays"][1], inst=inst,
                             scene=scene_aabb, dbg_cap=128)
    num = res[7].reshape(-1).astype(np.int64)
    assert np.array_equal(num, ora["hit_num"].reshape(-1).astype(np.int64))
    ids = res[6].view(np.uint32).reshape(num.size, -1)
    for r in range(num.size):
        k = min(int(num[r]), 128)
        assert np.array_equal(ids[r, :k], ora["hit_ids"][r, :k])


@pytest.mark.parametrize("sph_half,out_half", [(True, False), (False, True), (True, True)])
def test_fp16_feature_io_matches_the_fp32_path_and_the_oracle(sph_half, out_half):
    """render.particle_feature_half / feature_output_half (setup_3dgrt.py:41-44; optixTracer.cpp:52-60, 903-909): half coefficient
    buffer, half [H,W,3] feature image, fp32 arithmetic.  Half coefficients = the fp32 path on the rounded coefficients bit for bit;
    the half image = the fp32 image rounded once; the backward starts from the rounded image (referenceSlangBwdOptix.cu:116-117)."""
    import torch
    n, w, h = 800, 48, 32
    scene = _scene(n, w, h, 0.1)
    rng = np.random.default_rng(4)
    g_rad = rng.normal(size=(h, w, 3)).astype(np.float32)
    g_dns = rng.normal(size=(h, w, 1)).astype(np.float32)
    rounded = dict(scene, sph=oracle.round_to_half(scene["sph"])) if sph_half else scene
    ref32 = _render(rounded, g_rad, g_dns)
    gpu = _render(scene, g_rad, g_dns, particle_feature_half=sph_half, feature_output_half=out_half)
    f32, f = ref32["out"]["pred_features"], gpu["out"]["pred_features"]
    assert f.dtype == torch.float32   # always fp32 to the caller (threedgrt_tracer/tracer.py:98)
    assert torch.equal(f, f32.half().float() if out_half else f32)
    for k in ("pred_opacity", "pred_dist", "hits_count"):
        assert torch.equal(gpu["out"][k], ref32["out"][k]), k
    if not out_half:   # same hits, same values; the gradient atomics commute only up to rounding
        assert rel_err(gpu["grads"][0], ref32["grads"][0]) < 2e-5 and rel_err(gpu["grads"][1], ref32["grads"][1]) < 2e-5
    cfg = oracle.default_grt_config()
    ora = oracle.grt_forward(cfg, rounded["density12"], rounded["sph"], 3, 1e-3, scene["T"], *scene["rays"])
    got = f[0].detach().cpu().numpy()
    ulp = np.spacing(np.abs(got).astype(np.float16)).astype(np.float32) if out_half else 0.0
    assert (np.abs(got - ora["features"]) > 1e-4 + 0.5 * ulp).mean() <= 5e-3
    if out_half:
        ora = dict(ora, features=oracle.round_to_half(ora["features"]))
    rd, rs = oracle.grt_backward(cfg, 3, 1e-3, ora, g_rad, g_dns, np.zeros((h, w, 1), np.float32))
    n_flip = int((gpu["out"]["hits_count"][0, ..., 0].detach().cpu().numpy() != ora["hit_count"][..., 0]).sum())

    def trimmed(a, b, drop):
        e = np.abs(np.asarray(a, np.float64) - b).reshape(a.shape[0], -1).max(1)
        e = np.sort(e)[: max(1, len(e) - drop)]
        return float(e.max() / (np.abs(b).max() + 1e-12))

    gd, gs = gpu["grads"]
    for name, sl in {"position": slice(0, 3), "density": slice(3, 4), "rotation": slice(4, 8), "scale": slice(8, 11)}.items():
        assert trimmed(gd[:, sl], rd[:, sl], 3 * n_flip) < 1e-3, name
    assert trimmed(gs, rs, 3 * n_flip) < 1e-3 and gs.dtype == np.float32


def test_trim_releases_the_scratch_and_requires_a_new_bvh():
    import torch
    abi = importlib.import_module("3dgrut_amd._abi")
    scene = _scene(3000, 48, 32, 0.06)
    tr = _tracer()
    g = syn.SimpleGaussians(scene["density12"], scene["sph"])
    batch = torch_batch(scene["batch"], "cuda")
    tr.build_acc(g, rebuild=True)
    a = tr.render(g, batch)["pred_features"].detach().clone()
    live = abi.allocator_stats["live_bytes"]
    assert live > 0
    tr.tracer_wrapper.trim()
    assert abi.allocator_stats["live_bytes"] < live
    with pytest.raises(RuntimeError, match="build_bvh"):
        tr.render(g, batch)
    tr.build_acc(g, rebuild=True)
    assert torch.equal(tr.render(g, batch)["pred_features"].detach(), a)


NHT_CONF = {"feature_type": "nht", "nht_features": {"dim": 48, "activation": {"type": "sincos", "num_frequencies": 1}, "interpolation_type": "barycentric"}}


def _nht_tracer(**render_kw):
    gt = importlib.import_module("3dgrut_amd.grt_tracer")
    return gt.Tracer({"render": dict({"pipeline_type": "referenceSlang"}, **render_kw), "model": NHT_CONF})


@pytest.mark.parametrize("half", [False, True])
def test_nht_forward_matches_oracle(half):
    """model.feature_type = nht on the Slang pipeline (referenceSlangOptix.cu): [1,H,W,24] ray features against orc_grt_trace_nht_fwd
    fed the GPU's proxies (same hit sequences: the trace kernel is the SH one, the features are integrated from its hit log)."""
    import torch
    n, w, h = 3000, 64, 40
    scene = _scene(n, w, h, 0.06)
    feats = np.random.default_rng(12).uniform(-np.pi / 2, np.pi / 2, size=(n, 48)).astype(np.float32)
    tr = _nht_tracer(**(dict(particle_feature_half=True, feature_output_half=True) if half else {}))
    g = syn.SimpleGaussians(scene["density12"], feats, requires_grad=False)
    tr.build_acc(g, rebuild=True)
    with torch.no_grad():
        out = tr.render(g, torch_batch(scene["batch"], "cuda"))
    nat = tr.tracer_wrapper
    inst = nat.instances(n, "cuda").cpu().numpy()
    aabb = np.array(list(nat.stats().scene_aabb), np.float32)
    ofeats = oracle.round_to_half(feats) if half else feats
    ora = oracle.grt_forward_nht(oracle.default_grt_config(), scene["density12"], ofeats, 1e-3, scene["T"], *scene["rays"], inst=inst, scene=aabb)
    f = out["pred_features"][0].cpu().numpy()
    assert f.shape == (h, w, 24) and out["pred_features"].dtype == torch.float32
    ulp = np.spacing(np.abs(f).astype(np.float16)).astype(np.float32) if half else 0.0
    flips = (out["hits_count"][0].cpu().numpy() != ora["hit_count"])[..., 0]
    bad = (np.abs(f - ora["features"]) > 1e-4 + 0.5 * ulp).any(-1) | (np.abs(out["pred_opacity"][0].cpu().numpy() - ora["density"])[..., 0] > 1e-4)
    assert flips.mean() <= 5e-3 and (bad & ~flips).mean() <= 2e-3, f"{bad.sum()} pixels beyond tolerance, {flips.sum()} flips"
    assert np.abs(f).max() > 0.3


@pytest.mark.parametrize("replay,half", [(True, False), (False, False), (True, True)])
def test_nht_backward_matches_oracle(replay, half):
    """The Slang backward pipeline with neural harmonic features (referenceSlangBwdOptix.cu) — by replay of the forward's log (default)
    and by traversal (render.backward_hit_replay: false) — against orc_grt_trace_nht_bwd: particle rows and feature buffer, 1e-3."""
    import torch
    n, w, h = 2500, 56, 40
    scene = _scene(n, w, h, 0.07)
    feats = np.random.default_rng(21).uniform(-np.pi / 2, np.pi / 2, size=(n, 48)).astype(np.float32)
    kw = dict(particle_feature_half=True, feature_output_half=True) if half else {}
    tr = _nht_tracer(backward_hit_replay=replay, **kw)
    g = syn.SimpleGaussians(scene["density12"], feats)
    tr.build_acc(g, rebuild=True)
    out = tr.render(g, torch_batch(scene["batch"], "cuda"), train=True)
    rng = np.random.default_rng(4)
    g_f = rng.normal(size=(h, w, 24)).astype(np.float32)
    g_d = rng.normal(size=(h, w, 1)).astype(np.float32)
    g_h = (rng.normal(size=(h, w, 1)) * 0.1).astype(np.float32)
    loss = (out["pred_features"][0] * torch.as_tensor(g_f, device="cuda")).sum() + (out["pred_opacity"][0] * torch.as_tensor(g_d, device="cuda")).sum() + \
           (out["pred_dist"][0] * torch.as_tensor(g_h, device="cuda")).sum()
    loss.backward()
    gd, gf = g.grads_packed()
    nat = tr.tracer_wrapper
    inst = nat.instances(n, "cuda").cpu().numpy()
    aabb = np.array(list(nat.stats().scene_aabb), np.float32)
    cfg = oracle.default_grt_config()
    ofeats = oracle.round_to_half(feats) if half else feats
    ora = oracle.grt_forward_nht(cfg, scene["density12"], ofeats, 1e-3, scene["T"], *scene["rays"], inst=inst, scene=aabb)
    n_flip = int((out["hits_count"][0, ..., 0].detach().cpu().numpy() != ora["hit_count"][..., 0]).sum())
    if half:
        ora = dict(ora, features=oracle.round_to_half(ora["features"]))
    rd, rf = oracle.grt_backward_nht(cfg, 1e-3, ora, g_f, g_d, g_h.reshape(-1))

    def trimmed(a, b, drop):
        e = np.abs(np.asarray(a, np.float64) - b).reshape(a.shape[0], -1).max(1)
        e = np.sort(e)[: max(1, len(e) - drop)]
        return float(e.max() / (np.abs(b).max() + 1e-12))

    for name, sl in {"position": slice(0, 3), "density": slice(3, 4), "rotation": slice(4, 8), "scale": slice(8, 11)}.items():
        assert trimmed(gd[:, sl], rd[:, sl], 3 * n_flip) < 1e-3, (name, trimmed(gd[:, sl], rd[:, sl], 3 * n_flip))
    assert trimmed(gf, rf, 3 * n_flip) < 1e-3 and gf.shape == (n, 48) and np.abs(rf).max() > 0


@pytest.mark.parametrize("prim,code", [("icosahedron", 1), ("trihexa", 7), ("sphere", 8), ("custom", 5)])
def test_nht_on_icosahedron_proxies_matches_reference_slang_programs_golden_and_the_oracle(prim, code):
    """(round 6: also trihexa and sphere - several proxies per particle, the feature kernels map the log's proxy ids to particles.)
    model.feature_type = nht with render.primitive_type = icosahedron (round 5; refused until then): the forward DIRECTLY against
    tests/golden/grt_trace_nht_mesh.npz = the reference's Slang forward programs compiled for MOGTracingIcosaHedron over the emulated OptiX's
    triangles (tests/test_oracle_cpu.py pins the oracle on the same file), then forward and both backward paths against the oracle given the
    GPU-built proxy records."""
    import os
    import sys
    import torch
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden
    gold = np.load(os.path.join(here, "golden", "grt_trace_nht_mesh.npz"))
    kw = make_golden.GRT_TRACE_SCENES[0]
    scene = make_scene(**kw)
    feats = gold[f"{prim}_s0_nht_features"]
    tr = _nht_tracer(primitive_type=prim)
    g = syn.SimpleGaussians(scene["density12"], feats, requires_grad=False)
    tr.build_acc(g, rebuild=True)
    with torch.no_grad():
        out = tr.render(g, torch_batch(scene["batch"], "cuda"))
    cnt = out["hits_count"][0].cpu().numpy()
    flips = (cnt != gold[f"{prim}_s0_hits_count"])[..., 0]
    assert flips.mean() <= 0.01, f"{int(flips.sum())} rays with a different number of accepted hits"
    e = np.abs(out["pred_features"][0].cpu().numpy() - gold[f"{prim}_s0_features"]).max(-1)
    tied = ~flips & (e > 1e-4)
    assert tied.mean() <= 0.02 and (not tied.any() or e[tied].max() < 5e-2), (int(tied.sum()), float(e.max()))
    ok = ~flips & ~tied
    assert np.abs(out["pred_opacity"][0].cpu().numpy() - gold[f"{prim}_s0_density"])[ok].max() < 1e-4
    assert np.abs(gold[f"{prim}_s0_features"]).max() > 0.5 and cnt.max() >= 15
    # against the oracle on a second scene, with gradients through both backward paths
    n, w, h = 2500, 56, 40
    scene = _scene(n, w, h, 0.07)
    feats = np.random.default_rng(21).uniform(-np.pi / 2, np.pi / 2, size=(n, 48)).astype(np.float32)
    cfg = oracle.default_grt_config(primitive_type=code)
    rng = np.random.default_rng(4)
    g_f = rng.normal(size=(h, w, 24)).astype(np.float32)
    g_d = rng.normal(size=(h, w, 1)).astype(np.float32)
    g_h = (rng.normal(size=(h, w, 1)) * 0.1).astype(np.float32)
    for replay in (True, False):
        tr = _nht_tracer(primitive_type=prim, backward_hit_replay=replay)
        g = syn.SimpleGaussians(scene["density12"], feats)
        tr.build_acc(g, rebuild=True)
        out = tr.render(g, torch_batch(scene["batch"], "cuda"), train=True)
        loss = (out["pred_features"][0] * torch.as_tensor(g_f, device="cuda")).sum() + (out["pred_opacity"][0] * torch.as_tensor(g_d, device="cuda")).sum() + \
               (out["pred_dist"][0] * torch.as_tensor(g_h, device="cuda")).sum()
        loss.backward()
        gd, gf = g.grads_packed()
        nat = tr.tracer_wrapper
        inst = nat.instances(n, "cuda").cpu().numpy()
        aabb = np.array(list(nat.stats().scene_aabb), np.float32)
        box_kw = dict(box8=nat.custom_boxes(n, "cuda").cpu().numpy()) if prim == "custom" else {}   # (the checker gets the GPU's world boxes like its records)
        ora = oracle.grt_forward_nht(cfg, scene["density12"], feats, 1e-3, scene["T"], *scene["rays"], inst=inst, scene=aabb, **box_kw)
        f = out["pred_features"][0].detach().cpu().numpy()
        flips = (out["hits_count"][0].detach().cpu().numpy() != ora["hit_count"])[..., 0]
        bad = (np.abs(f - ora["features"]) > 1e-4).any(-1) | (np.abs(out["pred_opacity"][0].detach().cpu().numpy() - ora["density"])[..., 0] > 1e-4)
        assert flips.mean() <= 5e-3 and (bad & ~flips).mean() <= 2e-3, (replay, int(bad.sum()), int(flips.sum()))
        rd, rf = oracle.grt_backward_nht(cfg, 1e-3, ora, g_f, g_d, g_h.reshape(-1))
        n_flip = int(flips.sum())

        def trimmed(a, b, drop):
            err = np.abs(np.asarray(a, np.float64) - b).reshape(a.shape[0], -1).max(1)
            err = np.sort(err)[: max(1, len(err) - drop)]
            return float(err.max() / (np.abs(b).max() + 1e-12))
        assert trimmed(gd[:, :11], rd[:, :11], 3 * n_flip) < 1e-3 and trimmed(gf, rf, 3 * n_flip) < 1e-3, (replay, trimmed(gd[:, :11], rd[:, :11], 3 * n_flip), trimmed(gf, rf, 3 * n_flip))


def test_custom_primitives_with_features_use_the_slang_pipelines_unsigned_hit_distance():
    """Round 6: render.primitive_type custom under the Slang pipelines.  particleDensityHitCustom (gaussianParticles.slang:489-523) reports
    canonicalRayDistance - a LENGTH - where the CUDA pipeline's intersectCustomParticle reports the distance with the ray parameter's sign: a
    particle whose point of maximum response lies BEHIND the ray origin (the origin inside its world box) is a candidate of the Slang pipeline
    and not of the CUDA one.  A camera in the middle of a few large particles: the feature frame (HIP and checker agree) counts hits the SH
    frame of the same proxies does not."""
    import torch
    n, w, h = 400, 32, 24
    scene = _scene(n, w, h, 0.9, max_density=0.5)
    feats = np.random.default_rng(2).uniform(-np.pi / 2, np.pi / 2, size=(n, 48)).astype(np.float32)
    tr = _nht_tracer(primitive_type="custom")
    g = syn.SimpleGaussians(scene["density12"], feats, requires_grad=False)
    tr.build_acc(g, rebuild=True)
    with torch.no_grad():
        out = tr.render(g, torch_batch(scene["batch"], "cuda"))
    nat = tr.tracer_wrapper
    inst = nat.instances(n, "cuda").cpu().numpy()
    aabb = np.array(list(nat.stats().scene_aabb), np.float32)
    box8 = nat.custom_boxes(n, "cuda").cpu().numpy()
    cfg = oracle.default_grt_config(primitive_type=5)
    ora = oracle.grt_forward_nht(cfg, scene["density12"], feats, 1e-3, scene["T"], *scene["rays"], inst=inst, scene=aabb, box8=box8)
    cnt = out["hits_count"][0].cpu().numpy()
    flips = (cnt != ora["hit_count"])[..., 0]
    bad = (np.abs(out["pred_features"][0].cpu().numpy() - ora["features"]) > 1e-4).any(-1)
    assert flips.mean() <= 5e-3 and (bad & ~flips).mean() <= 5e-3, (int(flips.sum()), int(bad.sum()))
    # the SH frame over the same proxies (signed distances): fewer processed hits
    sh = oracle.grt_forward(cfg, scene["density12"], scene["sph"], 3, 1e-3, scene["T"], *scene["rays"], inst=inst, scene=aabb, box8=box8)
    assert ora["hit_count"].sum() > sh["hit_count"].sum(), (float(ora["hit_count"].sum()), float(sh["hit_count"].sum()))


def test_nht_forward_matches_reference_slang_programs_golden():
    """The HIP 3DGRT path with neural harmonic features DIRECTLY against tests/golden/grt_trace_nht.npz — the reference's Slang forward
    pipeline (referenceSlangOptix.cu) run on the host over the emulated traversal: accepted-hit counts, visibility, 24 ray features."""
    import os
    import sys
    import torch
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden
    gold = np.load(os.path.join(here, "golden", "grt_trace_nht.npz"))
    for k, kw in enumerate(make_golden.GRT_TRACE_SCENES):
        scene = make_scene(**kw)
        tr = _nht_tracer()
        g = syn.SimpleGaussians(scene["density12"], gold[f"s{k}_nht_features"], requires_grad=False)
        tr.build_acc(g, rebuild=True)
        with torch.no_grad():
            out = tr.render(g, torch_batch(scene["batch"], "cuda"))
        cnt = out["hits_count"][0].cpu().numpy()
        flips = cnt != gold[f"s{k}_hits_count"]
        assert flips.mean() <= 0.01, f"scene {k}: {int(flips.sum())} rays with a different number of accepted hits"
        ok = ~flips[..., 0]
        assert np.abs(out["pred_features"][0].cpu().numpy() - gold[f"s{k}_features"])[ok].max() < 1e-4
        assert np.abs(out["pred_opacity"][0].cpu().numpy() - gold[f"s{k}_density"])[ok].max() < 1e-4
        hd = gold[f"s{k}_hit_distance"]
        assert np.abs(out["pred_dist"][0].cpu().numpy() - hd[..., :1])[ok].max() <= 1e-4 * max(1.0, np.abs(hd).max())
        vis = out["mog_visibility"].view(-1).view(torch.int32).cpu().numpy() != 0
        assert (vis != (gold[f"s{k}_visibility"] != 0)).sum() <= 3 * int(flips.sum())

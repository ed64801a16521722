"""GPU parity of the hybrid mesh + Gaussian path tracer (BASELINE config 5) — `playground_tracer.Tracer.render_playground`
over grt_build_mesh_bvh / grt_trace_hybrid — against

  * tests/golden/playground.npz: the reference's own programs (playgroundKernel.cu, materials.cuh, trace.cuh, 3dgrtTracer.cuh) compiled on
    the host over an emulated OptiX (oracle/ref/ref_playground.cpp): mirror / glass / textured diffuse, five PBR materials, smooth and
    hard normals, textures on / off, with and without Gaussians;
  * the C oracle (oracle/grt_oracle.c: orc_grt_hybrid_trace, itself pinned by that golden: tests/test_hybrid_oracle_cpu.py) on larger
    scenes, and at BASELINE config 5's size — 2 M Gaussians + a PBR mesh, fisheye rays, 1920x1080 — on a sample of the frame's rays.

Bar: mirror-bounce counts identical, radiance / opacity / last ray within 1e-4, outside a bounded set of threshold flips (a ray that
grazes a triangle edge, an accept test or a termination threshold within rounding: its PATH differs, not its arithmetic)."""
import importlib
import os

import numpy as np
import pytest

import oracle
import playground_scenes as ps

pytestmark = pytest.mark.gpu
syn = importlib.import_module("workloads.synthetic")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "playground.npz")


def _render(sc, opts, bounces, frame, tracer=None):
    """One frame of a playground scene through the plugin surface (materials as dicts with the reference's attribute names)."""
    import torch
    pt = importlib.import_module("3dgrut_amd.playground_tracer")
    tr = tracer or pt.Tracer({"render": {}})
    g = syn.SimpleGaussians(sc["density12"], sc["sph"], requires_grad=False)
    tr.build_gs_acc(g, rebuild=True)
    t = lambda a: None if a is None else torch.as_tensor(a, device="cuda")
    m = sc["mesh"]
    tr.build_mesh_acc(t(m["vertices"]), t(m["triangles"]))
    mats = [dict(diffuse_map=t(x["diffuse_tex"]), emissive_map=t(x["emissive_tex"]), metallic_roughness_map=t(x["metallic_roughness_tex"]),
                 normal_map=t(x["normal_tex"]), diffuse_factor=x["diffuse_factor"], emissive_factor=x["emissive_factor"],
                 metallic_factor=x["metallic_factor"], roughness_factor=x["roughness_factor"], alpha_mode=x["alpha_mode"],
                 alpha_cutoff=x["alpha_cutoff"], transmission_factor=x["transmission_factor"], ior=x["ior"]) for x in sc["materials"]]
    out = tr.render_playground(g, t(sc["ray_o"])[None], t(sc["ray_d"])[None], opts, t(m["triangles"]), t(m["vertex_normals"]), t(m["vertex_tangents"]),
                               t(m["vertex_has_tangents"]), t(m["prim_type"]), frame_id=frame, ray_max_t=t(sc["ray_max_t"])[None],
                               material_uv=t(m["mat_uv"]), material_id=t(m["mat_id"])[:, None], materials=mats, refractive_index=t(m["refractive_index"]),
                               envmap=t(sc["envmap"]), envmap_offset=t(sc["envmap_offset"]), max_pbr_bounces=bounces)
    torch.cuda.synchronize()
    res = {k: v[0].cpu().numpy() for k, v in out.items() if hasattr(v, "cpu")}
    return res, tr


def _errors(res, rgb, alpha, last_o, last_d):
    e = np.maximum(np.abs(res["pred_features"] - rgb).max(-1), np.abs(res["pred_opacity"][..., 0] - alpha))
    el = np.maximum(np.abs(res["last_ray_o"] - last_o).max(-1), np.abs(res["last_ray_d"] - last_d).max(-1))
    return e, el


@pytest.mark.parametrize("name,kind,opts,bounces,frame", ps.GOLDEN_CASES)
def test_hybrid_trace_matches_reference_programs_golden(name, kind, opts, bounces, frame):
    g = np.load(GOLDEN)
    sc = ps.make_playground_scene(kind)
    res, _ = _render(sc, opts, bounces, frame)
    e, el = _errors(res, g[f"{name}_rgb"], g[f"{name}_alpha"][..., 0], g[f"{name}_last_o"], g[f"{name}_last_d"])
    bad = (e > 1e-4) | (el > 1e-4)
    print(f"{name}: {int(bad.sum())} of {bad.size} rays beyond 1e-4; median err {np.median(e):.1e}, max {e.max():.2e}")
    assert bad.mean() <= 6e-3, (name, int(bad.sum()), float(e.max()))
    assert np.median(e) < 1e-5
    assert not np.any(res["pred_dist"]) and not np.any(res["pred_normals"]) and not np.any(res["hits_count"])   # never written by the reference's raygen


@pytest.mark.parametrize("kind,n,w,h,opts,bounces", [("mixed", 12000, 96, 64, ps.OPT_SMOOTH_NORMALS, 6), ("classic", 6000, 80, 48, 0, 7),
                                                     ("pbr", 0, 64, 40, ps.OPT_SMOOTH_NORMALS | ps.OPT_NO_TEXTURES, 5)])
def test_hybrid_trace_matches_oracle(kind, n, w, h, opts, bounces):
    sc = ps.make_playground_scene(kind, width=w, height=h, n=max(n, 1), seed=11)
    if n == 0:
        sc["density12"], sc["sph"] = sc["density12"][:0], sc["sph"][:0]
    res, tr = _render(sc, opts, bounces, 4)
    nat = tr.tracer_wrapper
    inst = nat.instances(n, "cuda").cpu().numpy() if n else np.zeros((0, 12), np.float32)
    scene_aabb = np.array(list(nat.stats().scene_aabb), np.float32)
    ora = oracle.grt_hybrid(oracle.default_grt_config(), sc["density12"], sc["sph"], 3, tr._min_transmittance, np.eye(4, dtype=np.float32), sc["ray_o"],
                            sc["ray_d"], sc["mesh"], opts=opts, max_pbr_bounces=bounces, materials=sc["materials"], envmap=sc["envmap"],
                            envmap_offset=sc["envmap_offset"], frame_number=4, inst=inst, scene=scene_aabb, ray_max_t=sc["ray_max_t"])
    e, el = _errors(res, ora["rgba"][..., :3], ora["rgba"][..., 3], ora["last_ray"][..., :3], ora["last_ray"][..., 3:])
    b = res["mirror_bounces"][..., 0]
    flips = (b != ora["bounces"].astype(np.int32)) | (e > 1e-4) | (el > 1e-4)
    print(f"{kind}: {int(flips.sum())} of {flips.size} rays differ (bounce count or > 1e-4); median err {np.median(e):.1e}")
    assert flips.mean() <= 6e-3, (int(flips.sum()), float(e.max()))
    if kind != "pbr":
        assert b.max() >= 1 and (b == 0).any()                       # some rays bounce off the mirror, some do not


def test_hybrid_trace_with_trihexa_proxies_matches_oracle():
    """Round 6 (advisor): render.primitive_type trihexa builds the tree over 3 N proxies; the hybrid tracer's segments must shade particle
    proxy / 3 (trace_segment indexed the parameter rows by the raw proxy: out of bounds beyond N, the wrong particle everywhere else)."""
    import torch
    pt = importlib.import_module("3dgrut_amd.playground_tracer")
    n, w, h = 5000, 72, 48
    sc = ps.make_playground_scene("classic", width=w, height=h, n=n, seed=5)
    tr = pt.Tracer({"render": {"primitive_type": "trihexa"}})
    res, tr = _render(sc, 0, 5, 2, tracer=tr)
    nat = tr.tracer_wrapper
    inst = nat.instances(n, "cuda").cpu().numpy()
    scene_aabb = np.array(list(nat.stats().scene_aabb), np.float32)
    ora = oracle.grt_hybrid(oracle.default_grt_config(primitive_type=7), sc["density12"], sc["sph"], 3, tr._min_transmittance, np.eye(4, dtype=np.float32),
                            sc["ray_o"], sc["ray_d"], sc["mesh"], opts=0, max_pbr_bounces=5, materials=sc["materials"], envmap=sc["envmap"],
                            envmap_offset=sc["envmap_offset"], frame_number=2, inst=inst, scene=scene_aabb, ray_max_t=sc["ray_max_t"])
    e, el = _errors(res, ora["rgba"][..., :3], ora["rgba"][..., 3], ora["last_ray"][..., :3], ora["last_ray"][..., 3:])
    flips = (res["mirror_bounces"][..., 0] != ora["bounces"].astype(np.int32)) | (e > 1e-4) | (el > 1e-4)
    print(f"trihexa hybrid: {int(flips.sum())} of {flips.size} rays differ; median err {np.median(e):.1e}")
    assert flips.mean() <= 6e-3 and np.isfinite(res["pred_features"]).all(), (int(flips.sum()), float(e.max()))
    assert res["pred_opacity"].max() > 0.3   # the Gaussians are seen


def test_primary_segment_through_packet_lists_equals_the_tree_walk(monkeypatch):
    """The first segment of every path starts at the camera: it scans the frame's packet lists (3DGRT forward) up to the
    surface the ray hits; bounced rays walk the tree.  Same candidates, same order: every output is identical to the all-walk run."""
    sc = ps.make_playground_scene("mixed", width=88, height=60, n=9000, seed=5)
    monkeypatch.delenv("GRUT_GRT_NO_LISTS", raising=False)
    a, tr = _render(sc, ps.OPT_SMOOTH_NORMALS, 5, 1)
    n_lists = int(tr.tracer_wrapper.stats().list_entries)
    monkeypatch.setenv("GRUT_GRT_NO_LISTS", "1")
    b, tr2 = _render(sc, ps.OPT_SMOOTH_NORMALS, 5, 1)
    assert n_lists > 0 and int(tr2.tracer_wrapper.stats().list_entries) == 0
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_mesh_refit_equals_rebuild():
    """build_mesh_acc(rebuild=False, allow_update=True) keeps the tree and refits its boxes (hybridTracer.h:121-122); the frame must be the
    one a full rebuild gives."""
    import torch
    sc = ps.make_playground_scene("classic", width=64, height=40, n=3000, seed=3)
    a, tr = _render(sc, 0, 7, 0)
    moved = dict(sc, mesh=dict(sc["mesh"], vertices=(sc["mesh"]["vertices"] + np.float32(0.03)).astype(np.float32)))
    fresh, _ = _render(moved, 0, 7, 0)
    v = torch.as_tensor(moved["mesh"]["vertices"], device="cuda")
    pt_tr = tr
    pt_tr.build_mesh_acc(v, torch.as_tensor(sc["mesh"]["triangles"], device="cuda"), rebuild=False, allow_update=True)
    # _render rebuilds the mesh tree itself: call the trace directly through the same tracer instead
    t = lambda x: None if x is None else torch.as_tensor(x, device="cuda")
    m = moved["mesh"]
    g = syn.SimpleGaussians(sc["density12"], sc["sph"], requires_grad=False)
    mats = [dict(diffuse_map=t(x["diffuse_tex"]), diffuse_factor=x["diffuse_factor"]) for x in sc["materials"]]
    out = pt_tr.render_playground(g, t(sc["ray_o"])[None], t(sc["ray_d"])[None], 0, t(m["triangles"]), t(m["vertex_normals"]), None, None, t(m["prim_type"]),
                                  material_uv=t(m["mat_uv"]), material_id=t(m["mat_id"])[:, None], materials=mats, refractive_index=t(m["refractive_index"]),
                                  envmap=t(sc["envmap"]), envmap_offset=t(sc["envmap_offset"]), max_pbr_bounces=7)
    torch.cuda.synchronize()
    assert np.array_equal(out["pred_features"][0].cpu().numpy(), fresh["pred_features"])
    assert np.array_equal(out["mirror_bounces"][0].cpu().numpy(), fresh["mirror_bounces"])


def test_config5_size_frame_matches_oracle_on_a_ray_sample():
    """BASELINE config 5: 2 M Gaussians + a PBR / mirror / glass / diffuse mesh, fisheye camera, 1920x1080.  The oracle tests every particle
    against every path segment, so it traces a sample of the frame's rays (their launch coordinates seed the random streams)."""
    import torch
    bench = importlib.import_module("bench")
    W, H, n = 1920, 1080, 2_000_000
    sc = bench.hybrid_scene(n, W, H, 0.008)
    res, tr = _render(sc, ps.OPT_SMOOTH_NORMALS, 7, 2)
    nat = tr.tracer_wrapper
    inst = nat.instances(n, "cuda").cpu().numpy()
    scene_aabb = np.array(list(nat.stats().scene_aabb), np.float32)
    sel = np.arange(0, W * H, 997)
    xy = np.stack([sel % W, sel // W], -1).astype(np.uint32)
    pick = lambda a: np.ascontiguousarray(a.reshape(W * H, -1)[sel][None])
    ora = oracle.grt_hybrid(oracle.default_grt_config(), sc["density12"], sc["sph"], 3, tr._min_transmittance, np.eye(4, dtype=np.float32),
                            pick(sc["ray_o"]), pick(sc["ray_d"]), sc["mesh"], opts=ps.OPT_SMOOTH_NORMALS, max_pbr_bounces=7, materials=sc["materials"],
                            envmap=sc["envmap"], envmap_offset=sc["envmap_offset"], frame_number=2, inst=inst, scene=scene_aabb,
                            ray_max_t=pick(sc["ray_max_t"])[..., 0], pixel_xy=xy[None], launch_width=W)
    sub = {k: v.reshape(W * H, -1)[sel][None] for k, v in res.items()}
    e, el = _errors(dict(pred_features=sub["pred_features"], pred_opacity=sub["pred_opacity"], last_ray_o=sub["last_ray_o"], last_ray_d=sub["last_ray_d"]),
                    ora["rgba"][..., :3], ora["rgba"][..., 3], ora["last_ray"][..., :3], ora["last_ray"][..., 3:])
    b = sub["mirror_bounces"][..., 0]
    flips = (b != ora["bounces"].astype(np.int32)) | (e > 1e-4) | (el > 1e-4)
    print(f"config 5: {int(flips.sum())} of {flips.size} sampled rays differ; median err {np.median(e):.1e}; rays with a mirror bounce {float((b > 0).mean()):.3f}")
    # (round 6: NO ray of the sample may differ - 0 of 2 080 measured; the allowance of 1 % that stood here until round 5 had nothing to allow)
    assert sel.size >= 2000 and int(flips.sum()) == 0, (int(flips.sum()), float(e.max()))
    assert (b > 0).any() and float(sub["pred_opacity"].mean()) > 0.05

"""GPU, BASELINE.json's sizes: HIP <-> oracle parity of whole frames, and size-independent properties of the same code path.

Parity (tests/parity_util.py): every configuration of BASELINE.json that fits one GPU is rendered and differentiated by the HIP
path and by the C oracle — C1 100 k Gaussians @ 400x400, C2 1 M @ 800x800, the bench frame 1 M @ 1920x1080, C4's cloud 3 M @
1920x1080 (3DGUT); C3 100 k @ 400x400 on every ray and 1 M @ 800x800 on a 4 k-ray subsample (3DGRT; the oracle tests every particle
against every ray).  Every pixel / ray is compared; the only exemptions are IDENTIFIED threshold flips (the oracle reproduces the
GPU's pixel by taking at most three of its own borderline decisions the other way), bounded at 0.2 % of the pixels.

Properties: the integer work of the binning (ordering, tiling of the list, multiplicities), run-to-run bitwise reproducibility of
images AND gradients (the 3DGUT gradient path has no atomics), linearity of the backward in the upstream gradient, value ranges."""
import ctypes as C
import importlib

import numpy as np
import pytest

import parity_util as pu
from scenes import torch_batch

pytestmark = pytest.mark.gpu
N, W, H = 1_000_000, 1920, 1080


@pytest.mark.parametrize("name,n,w,h,median_scale,pose", [("c1_100k_400", 100_000, 400, 400, 0.01, "device"), ("c1_100k_400", 100_000, 400, 400, 0.01, "host"),
                                                          ("c2_1m_800", 1_000_000, 800, 800, 0.01, "device"),
                                                          ("c4_1m_1080p", 1_000_000, 1920, 1080, 0.01, "device"), ("c4_1m_1080p", 1_000_000, 1920, 1080, 0.01, "host"),
                                                          ("c4_3m_1080p", 3_000_000, 1920, 1080, 0.007, "device")])
def test_gut_frame_matches_oracle_at_baseline_size(name, n, w, h, median_scale, pose):
    """3DGUT forward + backward against the oracle, every pixel and every particle (BASELINE.json: RGB / depth within 1e-4,
    gradients within 1e-3 relative).  Stages and the exemption rule: tests/parity_util.py.  pose = "device": the camera-to-world
    matrix is a GPU tensor and the library derives the sensor pose on the device - the path bench.py times; "host": a CPU tensor, the
    plugin derives it in numpy like the reference's.  Both must give bit-identical depth keys (stage A)."""
    stats = pu.gut_full_parity(n, w, h, median_scale, log=print, device_pose=pose == "device")
    pu.record_full_parity(f"{name}_{pose}_pose", stats)   # (recorded first: a failing configuration's numbers travel back too)
    pu.assert_gut_full_parity(stats)


VARIANTS = {
    # sorted mode: the K nearest pending hits per pixel (gutKBufferRenderer.cuh:62-122, 331-350)
    "k16": dict(render={"splat": {"k_buffer_size": 16}}, oracle_cfg={"k_buffer_size": 16}),
    # quartic particle kernel (3dgrt.yaml's degree on the 3DGUT path) under an OpenCV fisheye camera with distortion
    "deg4_fisheye": dict(camera_model="fisheye", render={"particle_kernel_degree": 4}, oracle_cfg={"particle_kernel_degree": 4}),
    # PARTICLE_FEATURE_HALF + FEATURE_OUTPUT_HALF (setup_3dgut.py:60-61)
    "fp16_io": dict(half=True, render={"particle_feature_half": True, "feature_output_half": True}),
}


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_gut_non_default_configurations_meet_the_staged_method_at_baseline_size(variant):
    """The bench frame (1 M Gaussians, 1920x1080, device-pose path) in the non-default configurations, through the SAME staged method as
    the default one - binning integer-exact, every pixel beyond 1e-4 identified by the oracle's own borderline decisions (the sorted
    mode's k-buffer order included: parity_util._composite), masked gradients to 1e-3 - instead of the trimmed error on 30 k-particle
    scenes these configurations had in round 3."""
    stats = pu.gut_full_parity(N, W, H, 0.01, log=print, device_pose=True, end_to_end=False, variant=VARIANTS[variant])
    # sorted mode: the k-buffer orders hits by their fp32 hit distance.  Until round 4 the kernels and the checker evaluated that distance
    # in different operation orders, pairs of hits that tie to rounding popped in either order and 0.34 % of the frame needed an exemption
    # of its own.  Now the kernels evaluate it in the checker's order (csrc/gut_render.hip: oracle_order_hit_t): same bits, same order, the
    # DEFAULT limits (measured round 5: 0.055 % exempt, every pixel identified, no order swap among the toggles) - and nothing beyond 1e-2
    # outside the identified accept / termination flips.
    pu.record_full_parity(f"c4_1m_1080p_{variant}", stats)
    pu.assert_gut_full_parity(stats)
    if variant == "k16":   # (outside the pixels where a decision was IDENTIFIED as taken the other way; an order tie is not among the decisions any more)
        assert stats["B_max_rgb_err_outside_identified_flips"] < 1e-2, stats


def test_gut_nht_frame_matches_oracle_at_baseline_size():
    """model.feature_type = nht (pixel-pair sweeps, csrc/gut_render_nht.inl) at the bench size against the oracle's restated Slang feature
    model (orc_gut_render_nht_fwd / _bwd; restated: the generated Slang header is not in the checkout)."""
    stats = pu.gut_full_parity_nht(N, W, H, 0.01, log=print)
    pu.record_full_parity("c4_1m_1080p_nht", stats)
    pu.assert_gut_full_parity_nht(stats)


def test_gut_nht_behind_the_sorted_hit_buffer_matches_oracle_at_baseline_size():
    """Round 6: model.feature_type = nht with render.splat.k_buffer_size = 16 (refused until then) at the bench size: the checker composites the
    GPU's own tile lists through the sorted hit buffer (orc_gut_render_nht_fwd with k_buffer_size 16).  Pixels whose hit count differs are
    decision flips; among the others a pixel beyond 1e-4 is an order tie of two hits in the buffer or the rounding class - bounded in number
    here (the K = 0 feature frame and the SH sorted frame identify theirs pixel by pixel; this first version of the combination bounds them)."""
    import importlib
    import torch
    n, w, h = N, W, H
    inp = pu.make_frame_inputs(n, w, h, 0.01)
    feats = np.random.default_rng(43).uniform(-np.pi / 2, np.pi / 2, size=(n, 48)).astype(np.float32)
    gt = importlib.import_module("3dgrut_amd.gut_tracer")
    model = {"feature_type": "nht", "nht_features": {"dim": 48, "activation": {"type": "sincos", "num_frequencies": 1}, "interpolation_type": "barycentric"}}
    tracer = gt.Tracer({"render": {"enable_hitcounts": True, "splat": {"k_buffer_size": 16}}, "model": model})
    hip = pu.hip_forward(dict(inp, sph=feats), tracer=tracer, device_pose=True)
    cfg = pu.oracle.default_gut_config(k_buffer_size=16)
    ora = pu.oracle.gut_forward_nht(cfg, inp["cam"], inp["ps"], inp["pe"], inp["d12"], feats, *inp["rays"], lists=(hip["sorted_idx"], hip["tile_ranges"]))
    d_img = np.abs(hip["fd"] - ora["feat_density"]).max(-1)
    d_dist = np.abs(hip["dist"] - ora["hit_distance"])[..., 0]
    X = hip["cnt"] != ora["hit_count"][..., 0]
    bad = ((d_img > 1e-4) | (d_dist > 1e-4)) & ~X
    stats = dict(N=n, W=w, H=h, P=w * h, variant="nht_k16", B_flip_pixels=int(X.sum()), B_flip_frac=float(X.mean()), B_beyond_1e4_outside_flips=int(bad.sum()),
                 B_beyond_1e4_frac=float(bad.mean()), B_max_err_outside_flips=float(d_img[~X].max()), B_median_err=float(np.median(d_img)),
                 B_feature_abs_max=float(np.abs(ora["feat_density"][..., :24]).max()), hits_per_pixel=float(ora["hit_count"].mean()))
    print(stats)
    pu.record_full_parity("c4_1m_1080p_nht_k16", stats)
    assert stats["B_flip_frac"] <= 2e-3 and stats["B_beyond_1e4_frac"] <= 2e-3 and stats["B_feature_abs_max"] > 0.3, stats
    assert stats["B_max_err_outside_flips"] < 5e-2 and stats["B_median_err"] < 1e-5, stats


@pytest.mark.parametrize("name,n,w,h,median_scale,ray_stride,prim", [
    ("c3_grt_100k_400", 100_000, 400, 400, 0.01, 1, "instances"), ("c3_grt_1m_800", 1_000_000, 800, 800, 0.01, 149, "instances"),
    # the reference paper's own 3DGRT configuration (configs/paper/3dgrt/base_ours_reference.yaml:16) and the custom-primitive proxies at
    # BASELINE config 3's size, through the same stages (round 5; until then they were compared on <= 20 k-particle scenes only)
    # (every ray with gradients at 100 k particles on 200 x 200 rays: the checker's 20-plane clip of all pairs took 352 s of the suite's 726 s at 400 x 400)
    ("c3_grt_icosahedron_100k_200", 100_000, 200, 200, 0.01, 1, "icosahedron"), ("c3_grt_icosahedron_1m_800", 1_000_000, 800, 800, 0.01, 293, "icosahedron"),
    ("c3_grt_custom_1m_800", 1_000_000, 800, 800, 0.01, 149, "custom"),
    # the flat proxies (round 5): plane-crossing candidates, the surfel branches of the per-hit math; tree walk
    ("c3_grt_trisurfel_1m_800", 1_000_000, 800, 800, 0.01, 149, "trisurfel"),
    # three offers per particle (every rhombus a proxy of its own); packet lists over the 3 N rhombi since round 6
    ("c3_grt_trihexa_1m_800", 1_000_000, 800, 800, 0.01, 149, "trihexa"),
    # round 6: EVERY ray with gradients (stride 1 = HIP backward against the checker's backward of the same frame) for the three proxies that
    # had a gradient comparison at test size only
    ("c3_grt_custom_100k_200", 100_000, 200, 200, 0.01, 1, "custom"), ("c3_grt_trisurfel_100k_200", 100_000, 200, 200, 0.01, 1, "trisurfel"),
    ("c3_grt_trihexa_100k_200", 100_000, 200, 200, 0.01, 1, "trihexa"),
    # render.primitive_type sphere (round 6): two offers per particle (entry and exit of the enclosing sphere), each a proxy of its own
    ("c3_grt_sphere_1m_800", 1_000_000, 800, 800, 0.01, 149, "sphere"), ("c3_grt_sphere_100k_200", 100_000, 200, 200, 0.01, 1, "sphere"),
    # render.pipeline_type barycentricSurfels over the trisurfel proxies (round 6; forward only: proxies, order and images)
    ("c3_grt_bary_1m_800", 1_000_000, 800, 800, 0.01, 149, "trisurfel+barycentricSurfels")])
def test_grt_frame_matches_oracle_at_baseline_size(name, n, w, h, median_scale, ray_stride, prim):
    """3DGRT (LBVH + software traversal) against the oracle: the per-ray order of processed particles bit-exact, images within
    1e-4, gradients within 1e-3 relative (full frame at 100 k particles; a 4 k-ray subsample of the 1 M / 800x800 frame)."""
    # 1 M particles: every 149th ray through all pairs (4296 rays; icosahedron, whose all-pairs test clips 20 planes per particle: every 293rd,
    # 2185 rays - the suite had grown to 14 minutes), then every 9th ray (71 k) with the oracle's scan restricted to the packet lists the GPU
    # built - checked to change nothing on the all-pairs sample (trihexa too since the end of round 6: its wide sample used to go through all pairs)
    grt = importlib.import_module("3dgrut_amd.grt_tracer")
    has_lists = True   # (custom and trihexa since round 6)
    prim, _, pipeline = prim.partition("+")
    stats = pu.grt_full_parity(n, w, h, median_scale, ray_stride=ray_stride, log=print, wide_stride=9 if ray_stride > 1 and has_lists else 0,
                               primitive_type=prim, pipeline_type=pipeline or None)
    pu.record_full_parity(name, stats)
    pu.assert_grt_full_parity(stats)


@pytest.mark.parametrize("name,prim,half", [("c3_grt_nht_1m_800", "instances", False), ("c3_grt_icosahedron_nht_1m_800", "icosahedron", False),
                                            ("c3_grt_nht_fp16_1m_800", "instances", True)])
def test_grt_feature_frames_match_oracle_on_a_ray_sample_at_baseline_size(name, prim, half):
    """Round 6: the configurations that were benched at 1 M Gaussians / 800x800 but compared at test size only - neural harmonic features
    on the 3DGRT plugin (instances and the paper's icosahedron proxies) and fp16 feature I/O - against the checker on every 149th ray of the
    BASELINE config 3 frame (4 296 rays, every particle offered to every one of them; the checker is fed the GPU's proxy records like the SH
    frames'): hit counts equal except on borderline accept decisions, ray features / opacity within 1e-4 (half output: + half an ulp of the
    half image) outside them."""
    stats = pu.grt_feature_parity(1_000_000, 800, 800, 0.01, ray_stride=149, primitive_type=prim, half=half, log=print)
    pu.record_full_parity(name, stats)
    assert stats["F_rays_compared"] >= 4000 and stats["F_feature_abs_max"] > 0.3
    assert stats["F_rays_hit_count_differs"] <= max(8, 5e-3 * stats["F_rays_compared"]), stats
    assert stats["F_rays_beyond_tolerance_without_a_flip"] == 0, stats
    assert stats["F_max_err_without_a_flip"] <= 1e-4 + stats["F_half_ulp_allowance"], stats


def _trimmed(got, ref, n_drop):
    """||got - ref||inf / ||ref||inf over rows after dropping the n_drop worst rows (3 per pixel whose hit count flipped)"""
    per = np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64)).reshape(len(ref), -1).max(1)
    if n_drop > 0:
        per = np.sort(per)[: max(1, len(per) - n_drop)]
    return float(per.max() / (np.abs(ref).max() + 1e-30))


@pytest.mark.parametrize("name", ["c4_1m_1080p", "c2_1m_800"])
def test_gut_frame_equals_the_reference_kernels_on_a_sample_at_baseline_size(name):
    """HIP = REFERENCE CODE at BASELINE size, on a sample: tests/golden/fullsize_gut_*.npz holds what the reference's own kernels
    (projectOnTiles over all 1 M particles, its binning, render on a 32 x 384 crop, renderBackward on the crop's first tile row) produced,
    compiled on the host from the sources where they lie (tests/golden/make_fullsize_golden.py).  Until round 4 the compiled reference only
    ever saw 700-particle scenes and the 1 M frames were compared with the C oracle alone."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"fullsize_gut_{name}.npz"))
    n, w, h, ms = int(g["n"]), int(g["W"]), int(g["H"]), float(g["median_scale"])
    inp = pu.make_frame_inputs(n, w, h, ms)
    hip = pu.hip_forward(inp, device_pose=True)
    # projection: a 4096-particle sample of projectOnTiles' outputs
    smp = g["sample"].astype(np.int64)
    tc_ref, tc = g["sample_tiles_count"], hip["tiles_count"][smp]
    assert int((tc != tc_ref).sum()) <= 2, "tile counts of the sampled particles"
    vis = (tc > 0) & (tc_ref > 0)
    assert vis.sum() > 1000
    assert np.array_equal(hip["depth"][smp][vis].view(np.uint32), g["sample_depth"][vis].view(np.uint32)), "depth keys are bit-identical to the reference kernel's"
    assert np.abs(hip["rgb"][smp][vis] - g["sample_features"][vis]).max() < 1e-5
    assert abs(int(hip["I"]) - int(g["num_entries"])) <= max(2, 1e-5 * int(g["num_entries"]))
    # binning: the sorted lists of the crop's tiles
    r0, c0, nr, nc = (int(v) for v in g["crop"])
    gx = (w + 15) // 16
    tiles = [(r0 + ty) * gx + (c0 + tx) for ty in range(nr) for tx in range(nc)]
    lens, at, differing = g["crop_list_lengths"], 0, 0
    for t, ln in zip(tiles, lens):
        a, b = hip["tile_ranges"][t]
        differing += not np.array_equal(hip["sorted_idx"][a:b], g["crop_lists"][at:at + int(ln)])
        at += int(ln)
    assert differing <= 1, f"{differing} of {len(tiles)} tile lists differ from the reference's"
    # render on the crop
    y0, x0, ch, cw = 16 * r0, 16 * c0, 16 * nr, 16 * nc
    fd, dist, cnt = hip["fd"][y0:y0 + ch, x0:x0 + cw], hip["dist"][y0:y0 + ch, x0:x0 + cw], hip["cnt"][y0:y0 + ch, x0:x0 + cw]
    flips = cnt != g["hit_count"][..., 0]
    bad = (np.abs(fd - g["feat_density"]).max(-1) > 1e-4) | (np.abs(dist - g["hit_distance"])[..., 0] > 1e-4)
    print(f"{name}: {int(flips.sum())} of {flips.size} crop pixels with another hit count, {int((bad & ~flips).sum())} beyond 1e-4 with the same count")
    assert flips.mean() <= 3e-3 and (bad & ~flips).sum() <= 2 and float(g["hit_count"].mean()) > 30
    # renderBackward on the crop's first tile row: density / rotation / scale rows and the per-particle radiance gradient (the position
    # rows additionally carry projectBackward's view-direction term here; the reference kernel under test is renderBackward alone)
    bh = int(g["bwd_rows"])
    g_full = np.zeros((h, w, 4), np.float32)
    g_full[y0:y0 + bh, x0:x0 + cw] = g["g_fd"]
    gd, gsph = pu.hip_backward(hip, g_full)
    touched = g["touched"].astype(np.int64)
    nflip = int(flips[:bh].sum())
    ref_d = g["grad_density"]
    for key, sl in (("density", slice(3, 4)), ("rotation", slice(4, 8)), ("scale", slice(8, 11))):
        assert _trimmed(gd[touched][:, sl], ref_d[:, sl], 3 * nflip) < 1e-3, key
    mask = hip["rgb"][touched] > 0       # projectBackward's clamp mask; SH band 0: d L / d coefficient = 0.2820948 * d L / d radiance
    got_rgb = gsph[touched][:, :3] / 0.28209479177387814
    assert _trimmed(got_rgb * mask, g["grad_features"] * mask, 3 * nflip) < 1e-3
    untouched = np.setdiff1d(np.flatnonzero(np.abs(gd[:, 3:11]).max(1) > 0), touched)
    assert len(untouched) <= 3 * nflip, "particles with a gradient that the reference's backward never touched"


@pytest.mark.parametrize("prim", ["instances", "icosahedron", "custom", "trisurfel", "trihexa", "sphere"])
def test_grt_frame_equals_the_reference_programs_on_a_ray_sample_at_baseline_size(prim):
    """BASELINE config 3's frame (1 M Gaussians, 800 x 800) against the reference's OWN 3DGRT programs - referenceOptix.cu and
    referenceBwdOptix.cu compiled on the host over the emulated traversal, every ray offered every one of the 1 M instances
    (tests/golden/fullsize_grt_c3_1m_800.npz, 1536 rays on a regular sub-grid): accepted-hit counts, images, last-hit distances, and the
    gradient rows of the particles those rays' backward touches, with the upstream gradient confined to the sampled rays.
    icosahedron (the paper's configuration) / custom / trisurfel / trihexa: the programs built for that primitive over the reference's own meshes / world boxes
    of all 1 M particles, 6144 rays, each ray offered a conservative superset of the particles it can touch
    (tests/golden/fullsize_grt_<prim>_c3_1m_800.npz, make_fullsize_golden.py: make_grt_prim)."""
    import os
    import sys
    import torch
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden as mg
    g = np.load(os.path.join(here, "golden", "fullsize_grt_c3_1m_800.npz" if prim == "instances" else f"fullsize_grt_{prim}_c3_1m_800.npz"))
    n, w, h, ms = int(g["n"]), int(g["W"]), int(g["H"]), float(g["median_scale"])
    syn = importlib.import_module("workloads.synthetic")
    grt = importlib.import_module("3dgrut_amd.grt_tracer")
    d12, sph = syn.cloud_trained_like(n, seed=42, median_scale=ms)
    K = syn.pinhole_intrinsics(w, h)
    ro, rd = syn.pinhole_rays(w, h, K)
    inp_T = syn.orbit_pose(0, n_views=8)
    batch = torch_batch(dict(rays_ori=ro, rays_dir=rd, T_to_world=inp_T[None], intrinsics=K), "cuda")
    tracer = grt.Tracer({"render": {"enable_hitcounts": True, "primitive_type": prim}})
    gs = syn.SimpleGaussians(d12, sph)
    tracer.build_acc(gs, rebuild=True)
    out = tracer.render(gs, batch, train=True)
    ys, xs = g["ys"].astype(np.int64), g["xs"].astype(np.int64)
    sh, sw = len(ys), len(xs)
    pick = lambda t: t[0].detach().cpu().numpy()[np.ix_(ys, xs)]
    cnt = pick(out["hits_count"])
    flips = (cnt != g["hits_count"])[..., 0]
    ok = ~flips
    print(f"{int(flips.sum())} of {flips.size} sampled rays with another number of accepted hits; hits per ray {float(g['hits_count'].mean()):.1f}")
    assert flips.mean() <= 5e-3 and float(g["hits_count"].mean()) > 20
    # same count, other value: two hits whose distances tie to rounding are processed in the other order (the reference's
    # intersectInstanceParticle and this library's candidate arithmetic round differently; OptiX itself leaves the order of equal
    # distances open) - a handful of rays among ~100 k hits, each moved by the product of two alphas
    e_f = np.abs(pick(out["pred_features"]) - g["features"]).max(-1)
    e_o = np.abs(pick(out["pred_opacity"]) - g["density"])[..., 0]
    e_d = np.abs(pick(out["pred_dist"]) - g["hit_distance"][..., :1])[..., 0]
    tied = ok & ((e_f > 1e-4) | (e_o > 1e-4) | (e_d > 1e-4 * max(1.0, float(np.abs(g["hit_distance"]).max()))))
    print(f"{prim}: {int(tied.sum())} of {tied.size} rays with the same count beyond 1e-4 (order ties): max colour difference {float(e_f[tied].max()) if tied.any() else 0.0:.2e}")
    pu.record_full_parity(f"ref_programs_{prim}_c3_1m_800", dict(rays=int(flips.size), count_flips=int(flips.sum()), order_ties=int(tied.sum()),
                                                                 max_rgb_err_in_ties=float(e_f[tied].max()) if tied.any() else 0.0,
                                                                 max_rgb_err_elsewhere=float(e_f[ok & ~tied].max()), hits_per_ray=float(g["hits_count"].mean())))
    assert tied.mean() <= 5e-3
    # round 6: every such ray IDENTIFIED as an order tie instead of fenced (it used to pass with up to 5e-2 on 0.5 % of the rays): one
    # transposition of two neighbouring hits of the GPU's own sequence reproduces the reference's colour, opacity and distance within 1e-4,
    # and the two hits' distances are within a few float32 steps of each other (pu.grt_identify_order_ties)
    if tied.any():
        nat = tracer.tracer_wrapper
        hit_cap = 512 if prim == "sphere" else 256   # (a sphere's particle is offered at both roots: rays of 300 processed hits)
        frame = nat.make_frame(0, 3, tracer._min_transmittance, n, h, w, batch.T_to_world)
        res = nat.trace(frame, torch.as_tensor(d12, device="cuda").contiguous(), torch.as_tensor(sph, device="cuda").contiguous(),
                        batch.rays_ori.contiguous(), batch.rays_dir.contiguous(), hit_capacity=hit_cap)
        ids_all, num_all = res[6], res[7].reshape(-1)
        inst = nat.instances(n, "cuda").cpu().numpy()
        scene_aabb = np.array(list(nat.stats().scene_aabb), np.float32)
        box8 = nat.custom_boxes(n, "cuda").cpu().numpy() if prim == "custom" else None
        cases = []
        for sy_, sx_ in zip(*np.nonzero(tied)):
            pix = int(ys[sy_]) * w + int(xs[sx_])
            k = int(num_all[pix])
            assert k <= hit_cap
            cases.append((ro.reshape(-1, 3)[pix], rd.reshape(-1, 3)[pix], ids_all[pix, :k].cpu().numpy().view(np.uint32),
                          (g["features"][sy_, sx_], g["density"][sy_, sx_, 0], g["hit_distance"][sy_, sx_, 0]),
                          (pick(out["pred_features"])[sy_, sx_], pick(out["pred_opacity"])[sy_, sx_, 0], pick(out["pred_dist"])[sy_, sx_, 0])))
        recs = pu.grt_identify_order_ties(prim, cases, d12, sph, inst, scene_aabb, box8, inp_T, tracer._min_transmittance)
        # round 6, last session: what that search leaves open is explained with the reference programs' OWN hit log where the golden carries one
        # (tests/golden/fullsize_grt_<prim>_c3_1m_800_hitlog.npz, make_fullsize_golden.py hitlog_<prim>): ties at a round's last slot, which no
        # reordering of the GPU's sequence reproduces because the reference then never sees one of the two hits
        log_path = os.path.join(here, "golden", f"fullsize_grt_{prim}_c3_1m_800_hitlog.npz")
        if os.path.exists(log_path):
            hl = np.load(log_path)
            row_of = {int(r_): q for q, r_ in enumerate(hl["rays"])}
            for q, ((sy_, sx_), r_) in enumerate(zip(zip(*np.nonzero(tied)), recs)):
                key = int(sy_) * sw + int(sx_)
                if not r_["identified"] and key in row_of:
                    row = row_of[key]
                    r2 = pu.grt_identify_with_reference_log(prim, cases[q], hl["ids"][row, :int(hl["num"][row])], hl["ts"][row, :int(hl["num"][row])], d12, sph, inst,
                                                            scene_aabb, box8, inp_T, tracer._min_transmittance)
                    print(f"{prim}: with the reference programs' hit log: {r2}")
                    if r2["identified"]:
                        recs[q] = dict(r_, identified=True, kind="log:" + r2["kind"], with_reference_log=r2)
        for (sy_, sx_), r_ in zip(zip(*np.nonzero(tied)), recs):
            r_["ray_key"] = int(sy_) * sw + int(sx_)   # (row of the golden's ray sample: what make_fullsize_golden.py's hit log is keyed by)
            print(f"{prim}: tie ray: {r_}")
        pu.record_full_parity(f"ref_programs_{prim}_c3_1m_800_ties", dict(
            rays=len(recs), ties=int(sum(r_["kind"] == "tie" and r_["identified"] for r_ in recs)), rounding=int(sum(r_["kind"] == "rounding" for r_ in recs)),
            with_reference_log=int(sum(str(r_["kind"]).startswith("log:") for r_ in recs)),
            unidentified=int(sum(not r_["identified"] for r_ in recs)),
            max_float_steps_between_reordered_hits=float(max([max(r_.get("float_steps_between_reordered_hits", [0.0])) for r_ in recs if r_["identified"]] + [0.0])),
            max_err_after_reordering=float(max([r_.get("err_after_reordering", 0.0) for r_ in recs if r_["identified"]] + [0.0])), records=recs))
        # nothing may stay unidentified (the 5e-2 fence of round 5 is gone).  The rays the reordering search leaves open - one of the custom frame,
        # one of the trisurfel frame, one of the sphere frame - are explained with the reference programs' own hit log: custom / trisurfel are
        # ties at a round's last slot (the reference never returns one of the two hits);
        # sphere: the ninth ray is a tie of the two LAST neighbours - the reference processes Y and stops on the transmittance threshold with X
        # returned but unprocessed, here X comes first (9.5 float steps apart): `tie_at_end` in pu.grt_identify_with_reference_log, seen in the
        # programs' log of that ray, tests/golden/fullsize_grt_sphere_c3_1m_800_hitlog.npz)
        bad = [r_ for r_ in recs if not r_["identified"]]
        assert len(bad) == 0, bad
    ok = ok & ~tied
    # backward: the upstream gradient lives on the sampled rays only
    g_rad, g_dns, g_hit = mg.grt_trace_upstream(sh, sw)
    def full(a):
        z = np.zeros((h, w, a.shape[-1]), np.float32)
        z[np.ix_(ys, xs)] = a
        return z
    loss = (out["pred_features"][0] * torch.as_tensor(full(g_rad), device="cuda")).sum() + (out["pred_opacity"][0] * torch.as_tensor(full(g_dns), device="cuda")).sum() \
        + (out["pred_dist"][0] * torch.as_tensor(full(g_hit), device="cuda")).sum()
    loss.backward()
    torch.cuda.synchronize()
    gd, gsph = gs.grads_packed()
    touched = g["touched"].astype(np.int64)
    nflip = int(flips.sum()) + int(tied.sum())
    rows = touched[g["grad_rows"].astype(np.int64)]     # the stored rows: the 1500 largest gradients + 2500 random touched particles
    assert _trimmed(gd[rows][:, :11], g["grad_density"][:, :11], 3 * nflip) < 1e-3
    assert _trimmed(gsph[rows], g["grad_sph"], 3 * nflip) < 1e-3
    extra = np.setdiff1d(np.flatnonzero(np.abs(gd[:, :11]).max(1) > 0), touched)
    assert len(extra) <= 3 * nflip + 2, f"{len(extra)} particles carry a gradient the reference's backward program never touched"
    vis = out["mog_visibility"].view(-1).view(torch.int32).cpu().numpy() != 0
    assert vis[g["visible"].astype(np.int64)].mean() > 0.999      # what the sampled rays' forward marked visible is marked visible by the full frame


@pytest.fixture(scope="module")
def frame():
    import torch
    syn = importlib.import_module("workloads.synthetic")
    gt = importlib.import_module("3dgrut_amd.gut_tracer")
    d12, sph = syn.cloud_trained_like(N, seed=42, median_scale=0.01)
    K = syn.pinhole_intrinsics(W, H)
    ro, rd = syn.pinhole_rays(W, H, K)
    batch = torch_batch(dict(rays_ori=ro, rays_dir=rd, T_to_world=syn.orbit_pose(0, n_views=8)[None], intrinsics=K), "cuda")
    tracer = gt.Tracer({"render": {"enable_hitcounts": True, "splat": {}}})
    r = np.random.default_rng(11)
    g_a = torch.as_tensor(r.normal(size=(1, H, W, 4)).astype(np.float32), device="cuda")
    g_b = torch.as_tensor(r.normal(size=(1, H, W, 4)).astype(np.float32), device="cuda")

    def run(g_fd):
        g = syn.SimpleGaussians(d12, sph)
        out = tracer.render(g, batch, train=True)
        torch.autograd.backward([out["pred_features"], out["pred_opacity"]], [g_fd[..., :3].contiguous(), g_fd[..., 3:].contiguous()])
        torch.cuda.synchronize()
        return out, [p.grad for p in g.parameters()]
    return dict(run=run, g_a=g_a, g_b=g_b, tracer=tracer)


def test_images_and_gradients_are_bitwise_reproducible_and_in_range(frame):
    import torch
    out1, gr1 = frame["run"](frame["g_a"])
    keep = {k: out1[k].detach().clone() for k in ("pred_features", "pred_opacity", "pred_dist", "hits_count")}
    out2, gr2 = frame["run"](frame["g_a"])
    for k, v in keep.items():
        assert torch.equal(v, out2[k].detach()), k
    for a, b in zip(gr1, gr2):
        assert torch.equal(a, b)
    opa, feat, dist, cnt = keep["pred_opacity"], keep["pred_features"], keep["pred_dist"], keep["hits_count"]
    assert float(opa.min()) >= 0.0 and float(opa.max()) <= 1.0 and float(feat.min()) >= 0.0 and float(dist.min()) >= 0.0
    assert torch.equal(cnt > 0, opa > 0)                      # a pixel is opaque somewhere iff something was composited
    assert float((opa > 0.5).float().mean()) > 0.2            # the view is not empty
    assert all(bool(torch.isfinite(g).all()) for g in gr1)
    assert float(gr1[0].abs().max()) > 0 and float(gr1[4].abs().max()) > 0


def test_backward_is_linear_in_the_upstream_gradient(frame):
    _, ga = frame["run"](frame["g_a"])
    ga = [g.double() for g in ga]
    _, gb = frame["run"](frame["g_b"])
    gb = [g.double() for g in gb]
    _, gc = frame["run"](frame["g_a"] + 2.0 * frame["g_b"])
    for a, b, c in zip(ga, gb, gc):
        want = a + 2.0 * b
        err = float((c.double() - want).abs().max()) / (float(want.abs().max()) + 1e-30)
        assert err < 2e-4, err   # fp32 rounding of sums over thousands of pixels (measured 5e-5); a non-linear term would show as O(1)


def test_binning_integer_work_at_full_size(frame):
    import torch
    frame["run"](frame["g_a"])
    nat = frame["tracer"].tracer_wrapper
    st = nat.stats()
    I, tiles = int(st.num_intersections), int(st.num_tiles)
    assert st.num_particles == N and tiles == ((W + 15) // 16) * ((H + 15) // 16) and I > 5_000_000
    tc = torch.zeros(N, dtype=torch.int32, device="cuda")
    depth = torch.zeros(N, dtype=torch.float32, device="cuda")
    sidx = torch.zeros(I, dtype=torch.int32, device="cuda")
    rng = torch.zeros((tiles, 2), dtype=torch.int32, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert nat.lib.gut_debug_fetch(nat.handle, stream, p(tc), None, None, None, p(depth), None, p(sidx), p(rng)) == 0
    torch.cuda.synchronize()
    tc, depth = tc.cpu().numpy().view(np.uint32), depth.cpu().numpy()
    sidx, rng = sidx.cpu().numpy().view(np.uint32), rng.cpu().numpy().view(np.uint32)
    assert int(tc.sum(dtype=np.uint64)) == I
    lens = (rng[:, 1].astype(np.int64) - rng[:, 0].astype(np.int64))
    assert lens.min() >= 0 and lens.sum() == I
    nz = lens > 0
    starts = rng[nz, 0]
    assert starts[0] == 0 and np.array_equal(starts[1:], rng[nz, 1][:-1])           # the ranges tile [0, I) in tile order
    assert np.array_equal(np.bincount(sidx, minlength=N).astype(np.uint32), tc)      # multiplicity of a particle = its tile count
    # inside every tile: strictly ascending (depth bits, particle index) — the reference's stable sort order
    keys = (depth.view(np.uint32).astype(np.uint64)[sidx] << np.uint64(32)) | sidx.astype(np.uint64)
    tile_of = np.repeat(np.arange(tiles), lens)
    same_tile = tile_of[1:] == tile_of[:-1]
    assert np.all(keys[1:][same_tile] > keys[:-1][same_tile])


def test_grt_full_size_forward_is_reproducible_and_backward_is_stable():
    """3DGRT at BASELINE config 3's size (1 M Gaussians, 800x800, BVH rebuilt): the forward is bitwise reproducible (hit order
    and compositing do not depend on the traversal order), the per-particle visibility is exactly "took part in a processed hit
    of some ray", and the gradients — float atomics, hence order-dependent — agree run to run within rounding."""
    import torch
    syn = importlib.import_module("workloads.synthetic")
    grt = importlib.import_module("3dgrut_amd.grt_tracer")
    n, w, h = 1_000_000, 800, 800
    d12, sph = syn.cloud_trained_like(n, seed=42, median_scale=0.01)
    K = syn.pinhole_intrinsics(w, h)
    ro, rd = syn.pinhole_rays(w, h, K)
    batch = torch_batch(dict(rays_ori=ro, rays_dir=rd, T_to_world=syn.orbit_pose(0, n_views=8)[None], intrinsics=K), "cuda")
    tracer = grt.Tracer({"render": {"enable_hitcounts": True}})
    g_rgb = torch.as_tensor(np.random.default_rng(3).normal(size=(1, h, w, 3)).astype(np.float32), device="cuda")

    def run():
        g = syn.SimpleGaussians(d12, sph)
        tracer.build_acc(g, rebuild=True)
        out = tracer.render(g, batch, train=True)
        (out["pred_features"] * g_rgb).sum().backward()
        torch.cuda.synchronize()
        outs = {k: out[k].detach().clone() for k in ("pred_features", "pred_opacity", "pred_dist", "hits_count", "mog_visibility")}
        return outs, [p.grad.clone() for p in g.parameters()]
    o1, g1 = run()
    o2, g2 = run()
    for k in o1:
        assert torch.equal(o1[k], o2[k]), k
    assert float(o1["pred_opacity"].min()) >= 0.0 and float(o1["pred_opacity"].max()) <= 1.0 + 1e-6
    assert torch.equal(o1["hits_count"] > 0, o1["pred_opacity"] > 0)
    assert 0.05 < float(o1["mog_visibility"].bool().float().mean()) < 0.95
    for a, b in zip(g1, g2):
        assert bool(torch.isfinite(a).all())
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()) + 1e-12
    # particles no ray of the FORWARD processed: the reference's backward program can still be offered one (its traces end at other
    # distances than the forward's, so a candidate whose box the ray had left at a forward round's tmin can pass a backward round's
    # box-exit test, referenceBwdOptix.cu:123-131) — rarely
    unseen = ~o1["mog_visibility"].bool().view(-1)
    with_grad = (g1[0].abs().amax(1) > 0) & unseen
    assert float(with_grad.float().sum()) <= 0.02 * float((~unseen).float().sum())


def test_grt_default_backward_equals_the_rederived_backward_at_full_size():
    """BASELINE config 3's frame (1 M Gaussians, 800x800): the default backward — the forward's log of processed hits and ghosts walked
    with the backward program's own trace intervals — against the backward that traverses again round by round
    (render.backward_hit_replay = false: the reference's program literally, checked against the oracle at 100 k particles where the
    oracle can afford every ray).  At this size 70 % of the rays lose a hit to the endT clip of the backward's traces and 8 % of them
    process another hit SET than the forward (oracle statistics); all of it must come out the same."""
    import torch
    syn = importlib.import_module("workloads.synthetic")
    grt = importlib.import_module("3dgrut_amd.grt_tracer")
    n, w, h = 1_000_000, 800, 800
    d12, sph = syn.cloud_trained_like(n, seed=42, median_scale=0.01)
    K = syn.pinhole_intrinsics(w, h)
    ro, rd = syn.pinhole_rays(w, h, K)
    batch = torch_batch(dict(rays_ori=ro, rays_dir=rd, T_to_world=syn.orbit_pose(0, n_views=8)[None], intrinsics=K), "cuda")
    rng = np.random.default_rng(5)
    g_rgb = torch.as_tensor(rng.normal(size=(1, h, w, 3)).astype(np.float32), device="cuda")
    g_opa = torch.as_tensor(rng.normal(size=(1, h, w, 1)).astype(np.float32), device="cuda")

    def grads(**kw):
        tracer = grt.Tracer({"render": dict(enable_hitcounts=True, **kw)})
        g = syn.SimpleGaussians(d12, sph)
        tracer.build_acc(g, rebuild=True)
        sig, cnt = tracer.tracer_wrapper.backward_signature(w * h, "cuda")
        out = tracer.render(g, batch, train=True)
        ((out["pred_features"] * g_rgb).sum() + (out["pred_opacity"] * g_opa).sum()).backward()
        torch.cuda.synchronize()
        return [p.grad.double() for p in g.parameters()], tracer.tracer_wrapper.stats(), sig.cpu().numpy(), cnt.cpu().numpy()
    (a, st, sig_a, cnt_a), (b, _, sig_b, cnt_b) = grads(), grads(backward_hit_replay=False)
    print(f"rays with re-derived rounds {st.bwd_rederived_rays}, ghost-premise failures {st.bwd_premise_rays}")
    assert st.bwd_premise_rays == 0 and st.bwd_rederived_rays <= 5e-2 * w * h
    # ray by ray: the SAME hits were differentiated (count and order-independent signature of the particle set)
    differs = (cnt_a != cnt_b) | (sig_a != sig_b)
    print(f"rays whose differentiated hit set differs: {int(differs.sum())} of {w * h}; hits per ray: mean {cnt_b.mean():.1f}, max {cnt_b.max()}")
    assert not differs.any(), [(int(r), int(cnt_a[r]), int(cnt_b[r])) for r in np.flatnonzero(differs)[:12]]
    for x, y in zip(a, b):
        err = float((x - y).abs().max()) / (float(y.abs().max()) + 1e-30)
        assert err < 1e-4, err   # same hits in the same traces; float atomics add in another order (measured ~1e-5)


def test_device_pose_path_is_the_host_pose_path_bit_for_bit_at_full_size():
    """The bench - and any trainer whose batch lives on the GPU - leaves the camera-to-world matrix on the device and the library
    derives the sensor pose there (GutFrame::device_T_to_world); the reference's plugin does it on the host in numpy / torch
    (tracer.py:359-423: float64 general inverse, one rounding to float32, float32 quaternion).  csrc/gut_poses.hip runs that very
    arithmetic on the GPU with contraction off, so the two paths must render THE SAME BITS on the bench frame: depth keys, lists,
    images, hit counts and every gradient (round 3 inverted in fp32 on the device: 0.2 % of the pixels moved)."""
    import torch
    syn = importlib.import_module("workloads.synthetic")
    inp = pu.make_frame_inputs(N, W, H, 0.01)
    g_fd_np, _ = syn.upstream_grads(W, H)
    g_fd = g_fd_np * (W * H)
    dev, host = pu.hip_forward(inp, device_pose=True), pu.hip_forward(inp, device_pose=False)
    assert np.array_equal(dev["depth"].view(np.uint32), host["depth"].view(np.uint32)), "depth keys differ between the pose paths"
    for k in ("tiles_count", "sorted_idx", "tile_ranges", "fd", "dist", "cnt", "vis", "rgb"):
        assert np.array_equal(dev[k], host[k]), k
    (gd_d, gs_d), (gd_h, gs_h) = pu.hip_backward(dev, g_fd), pu.hip_backward(host, g_fd)
    assert np.array_equal(gd_d, gd_h) and np.array_equal(gs_d, gs_h)
    assert float(np.abs(gd_h).max()) > 0

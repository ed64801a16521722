#!/usr/bin/env python
"""bench.py — train rays/s of the 3DGUT hot path (forward + backward of one view per GPU per step).

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched
under torch.distributed.run, one rank per GPU (RCCL).  Rank 0 prints ONE JSON line.

Workload: BASELINE.json's metric config — 1M synthetic "trained-like" Gaussians (SURVEY §8d cloud B), pinhole
camera with the Lego field of view on a radius-4 orbit, 1920x1080, SH degree 3, default 3dgut.yaml flags.
Each rank renders its own view (weak scaling); for N>1 the packed Gaussian gradients are sum-all-reduced
over RCCL every step (the path's one exchange step).
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver: RCCL needs it for the multi-rank runs

import numpy as np
import torch

WORKLOADS = {
    # name: (num_gaussians, width, height, median_scale)
    "c4_1m_1080p": (1_000_000, 1920, 1080, 0.01),
    "c2_1m_800": (1_000_000, 800, 800, 0.01),
    "c4_3m_1080p": (3_000_000, 1920, 1080, 0.007),  # BASELINE config 4's cloud size (one view per GPU)
    "c1_100k_400": (100_000, 400, 400, 0.01),
    # BASELINE config 3 (not the default bench line): 3DGRT software-BVH primary rays, forward + backward
    "c3_grt_1m_800": (1_000_000, 800, 800, 0.01),
    "c3_grt_100k_400": (100_000, 400, 400, 0.01),
    # BASELINE config 5: hybrid mesh + Gaussian path tracing (reflection / refraction), 2 M Gaussians + mesh, fisheye camera, 1080p, forward only
    "c5_hybrid_2m_1080p": (2_000_000, 1920, 1080, 0.008),
    # model.feature_type = nht at the headline size
    "c4_nht_1m_1080p": (1_000_000, 1920, 1080, 0.01),
    "c3_grt_nht_1m_800": (1_000_000, 800, 800, 0.01),
    # render.primitive_type = icosahedron (the reference paper's 3DGRT configuration, configs/paper/3dgrt/base_ours_reference.yaml:16): tree walk
    "c3_grt_icosa_1m_800": (1_000_000, 800, 800, 0.01),
    # render.primitive_type = custom (world boxes: tree walk every round) / trisurfel (flat proxies, on the packet lists)
    "c3_grt_custom_1m_800": (1_000_000, 800, 800, 0.01),
    "c3_grt_trisurfel_1m_800": (1_000_000, 800, 800, 0.01),
    "c3_grt_trihexa_1m_800": (1_000_000, 800, 800, 0.01),    # three rhombi per particle, each a proxy of its own
    "c3_grt_sphere_1m_800": (1_000_000, 800, 800, 0.01),     # OptiX's built-in spheres: entry and exit of every enclosing sphere, each a proxy of its own
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E peak


def byte_model(N, Nv, I, P, tile_bits, sph_coeffs=16):
    """Algorithmic bytes per launch of each stage (with this design's key widths)."""
    sh = sph_coeffs * 12
    return {
        "project": N * (48 + 4 + 56) + Nv * sh,
        "depth_sort": N * 4 * (4 + 16),                      # 4 passes x (hist read 4 B + pairs in/out 16 B)
        "scan": N * 8,
        "expand": N * 44 + I * 8,
        "tile_sort": I * ((tile_bits + 7) // 8) * (4 + 16),
        "tile_ranges": I * 4,
        "render_fwd": I * (4 + 48 + 12) + P * (24 + 24),
        "render_bwd": I * (4 + 48 + 12) + I * 112 + P * (24 + 16 + 4 + 16 + 4),
        "project_bwd": Nv * (12 + 48 + sh + sh + 12) + (N - Nv) * sh,
    }


def valu_fraction(stage, kernel_ms):
    """Share of the kernel's duration that its VALU instructions alone keep the SIMDs busy.  Dynamic count: rocprofv3 SQ_INSTS_VALU
    of the same launch (profiles/sq_insts_valu.json); cost per instruction: the kernel's static instruction mix priced with the
    wave64 issue times MEASURED on this chip (scripts/valu_calib.hip -> profiles/r02a_valu_calib.json; scripts/valu_model.py ->
    profiles/valu_model.json): v_mul 1.14 ns, v_fma 1.73 ns, v_pk_* 2.2 ns, v_exp / v_rcp 3.4 ns per instruction per SIMD at
    saturation — not the 2 cycles a datasheet suggests for every fp32 op."""
    try:
        insts = json.load(open(os.path.join(ROOT, "profiles", "sq_insts_valu.json")))[stage]
        model = json.load(open(os.path.join(ROOT, "profiles", "valu_model.json")))["kernels"][stage]
    except Exception:
        return None
    busy_ms = insts / 1024.0 * model["avg_ns_per_instruction"] * 1e-6
    return {"insts_valu": insts, "avg_ns_per_inst": model["avg_ns_per_instruction"], "mix": model["mix"], "busy_ms_per_simd": busy_ms,
            "frac": busy_ms / kernel_ms, "source": "profiles/sq_insts_valu.json x profiles/valu_model.json (profiles/r02a_valu_calib.json)"}


def touched_bytes(stage, st, P, I):
    """Bytes a sweep can actually touch: it stops at the rays' termination, so it evaluates E << I tile-list entries (counted on the
    device in one instrumented frame, GutStats.fwd/bwd_entries_*).  Per evaluated entry of a half-tile wave: 4 B list entry + 4 B
    particle index + 48 B particle row + 12 B radiance = 68 B (round 3's direct lists: 4 B list entry + one 64-B particle record, the same 68 B); per accepted entry of the gradient sweep: one 64-B slot + 1 flag byte;
    per pixel: rays 24 B + outputs 24 B (forward) or rays 24 B + image 16 B + upstream gradient 16 B + (checkpoint 40 B per pixel and
    256-entry segment started) in the gradient sweep."""
    if stage == "render_fwd" and st.fwd_entries_evaluated:
        return int(st.fwd_entries_evaluated) * 68 + P * 48 + (int(st.fwd_entries_evaluated) // 256) * 128 * 20
    if stage == "render_bwd" and st.bwd_entries_evaluated:
        return int(st.bwd_entries_evaluated) * 68 + int(st.bwd_entries_accepted) * 65 + P * 56 + (int(st.bwd_entries_evaluated) // 256) * 128 * 20 + 2 * I
    return None


def cpu_baseline(seconds_budget=20.0):
    """Oracle (CPU restatement of the reference) on a bounded sample: C1-sized frames, fwd+bwd."""
    import oracle
    from workloads.scenes import make_scene
    syn = importlib.import_module("workloads.synthetic")
    n, w, h, ms = WORKLOADS["c1_100k_400"]
    scene = make_scene(n=n, width=w, height=h, median_scale=ms)
    cfg = oracle.default_gut_config()
    g_fd, g_dist = syn.upstream_grads(w, h)
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    t0 = time.time()
    frames = 0
    while True:
        fwd = oracle.gut_forward(cfg, scene["cam"], scene["pose_start"], scene["pose_end"], 3, scene["density12"], scene["sph"], *scene["rays"])
        oracle.gut_backward(cfg, scene["cam"], 3, fwd, g_fd, g_dist)
        frames += 1
        if time.time() - t0 > seconds_budget * 0.5 or frames >= 8:
            break
    dt = time.time() - t0
    return {"value": frames * w * h / dt, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{frames} fwd+bwd frames of {n} Gaussians at {w}x{h} (config C1) through the C oracle, OpenMP over pixels"}


def grt_roofline(work, P, stages):
    """Forward trace against the HBM roofline.  Algorithmic bytes (SURVEY §8d restated for this design, per wave where the 8x8 ray packet
    shares a fetch): 64 B per node visit of a wave (tree walk) or per candidate test of a packet (48 B proxy record + 16 B bounds, packet
    lists); 40 B per list entry built (12 B written, the two sort passes, 4 B read back); 240 B per processed hit (48 B particle + 192 B
    SH, gathered per ray); 64 B per ray of inputs / outputs.  Neither HBM nor VALU binds this kernel: the fraction is
    reported as the contract asks, `traffic` is the measured HBM volume, `valu` the issue-slot fraction."""
    if not work or "forward_render" not in stages:
        return None
    byts = (work["wave_node_visits"] + work.get("packet_tests", 0)) * 64 + work.get("list_entries", 0) * 40 + work["processed_hits"] * 240 + P * 64
    ms = stages["forward_render"]
    achieved = byts / (ms * 1e-3) / 1e9
    traffic = None
    try:
        traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("c3_grt_1m_800", {}).get("trace_fwd")
    except Exception:
        pass
    r = {"bound": "hbm", "kernel": "grt_trace_fwd", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": traffic, "algorithmic_bytes": byts, "kernel_ms": ms,
         "note": "forward_render = packet-list build (cones, binning, sorts) + trace"}
    v = valu_fraction("grt_trace_fwd", ms)
    if v:
        r["valu"] = v
    return r


def atomic_calibration():
    """profiles/rNN_atomic_calib.json of the latest round that has one (scripts/atomic_calib.hip run on an MI355X): float words per second
    of the replay backward's own atomic-instruction shape ("rows59": 59 consecutive words of one row per instruction, 256 MB table)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_atomic_calib.json")), reverse=True):
        try:
            c = json.load(open(path))
            return {"file": os.path.relpath(path, ROOT), "rows59_words_per_s": float(c["rows59"]["words_per_s"]),
                    "rows59_instructions_per_s": float(c["rows59"]["instructions_per_s"]), "rows59_hot_words_per_s": float(c["rows59_hot"]["words_per_s"])}
        except Exception:
            continue
    return None


def grt_backward_roofline(work, P, stages, atomics=None):
    """Replay backward.  What binds it is the L2's atomic units, not HBM or VALU (22 % of the VALU issue time; the float words
    are merged in the L2 and never reach HBM one by one - the PMC passes see a third of the "algorithmic" bytes).  So it is priced against
    the unit that binds it: `achieved` = float words its atomic instructions carry per second (counted on the device by the kernel itself,
    GrtStats::bwd_atomic_words) against `peak` = the rate scripts/atomic_calib.hip measures on this chip for the same instruction shape
    (one instruction = up to 59 consecutive words of one particle's rows).  `frac_traffic` keeps the HBM view beside it: PMC bytes / time /
    8 TB/s.  `algorithmic_words` = 59 per differentiated hit, what the reference issues as 59 separate atomics; same-slot merging (lanes of
    a wave that meet the same particle leave as one instruction) is why fewer are issued."""
    if not work or "backward_render" not in stages:
        return None
    hits = work["processed_hits"]
    ms = stages["backward_render"]
    traffic = None
    try:
        traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("c3_grt_1m_800", {}).get("replay_bwd")
    except Exception:
        pass
    words = int(atomics["words"]) if atomics and atomics.get("words") else hits * 59
    cal = atomic_calibration()
    achieved = words / (ms * 1e-3)
    r = {"bound": "l2_atomic", "kernel": "grt_replay_bwd", "achieved": achieved, "peak": cal["rows59_words_per_s"] if cal else None, "unit": "float words/s",
         "frac": achieved / cal["rows59_words_per_s"] if cal else None, "calibration": cal, "kernel_ms": ms,
         "atomic_words_issued": words, "atomic_instructions_issued": int(atomics["instructions"]) if atomics else None,
         "algorithmic_words": hits * 59, "traffic": traffic,
         "frac_traffic": (traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
         "note": "bound by the L2 atomic units: words/s against the calibrated rate of the same instruction shape (scripts/atomic_calib.hip); "
                 "kernel_ms is the whole backward stage (replay + the re-derivation launch on the side stream)"}
    return r


def bench_grt(args, world, rank, dev, dist, n, W, H, ms, name=None, emit=True):
    """3DGRT: BVH build + forward + backward of one view per GPU per step (the reference rebuilds the BVH every iteration,
    trainer.py:1257-1263).  Traversal is latency / divergence bound; the line reports rays/s and per-stage ms.
    emit=False: return the result object instead of printing it (the `secondary` entry of the default bench line)."""
    name = name or args.workload
    syn = importlib.import_module("workloads.synthetic")
    grt = importlib.import_module("3dgrut_amd.grt_tracer")
    from workloads.scenes import torch_batch
    d12, sph = syn.cloud_trained_like(n, seed=42, median_scale=ms)
    K = syn.pinhole_intrinsics(W, H)
    ro, rd = syn.pinhole_rays(W, H, K)
    batch = torch_batch(dict(rays_ori=ro, rays_dir=rd, T_to_world=syn.orbit_pose(rank, n_views=max(world, 8))[None], intrinsics=K), dev)
    nht = "nht" in name   # model.feature_type = nht on the Slang pipelines (48 feature floats per particle -> 24 ray features)
    if nht:
        sph = np.random.default_rng(7).uniform(-np.pi / 2, np.pi / 2, size=(n, 48)).astype(np.float32)
        tracer = grt.Tracer({"render": {"enable_kernel_timings": True, "pipeline_type": "referenceSlang"},
                             "model": {"feature_type": "nht", "nht_features": {"dim": 48, "activation": {"type": "sincos", "num_frequencies": 1},
                                                                               "interpolation_type": "barycentric"}}})
    else:
        prim = next((v for k, v in (("icosa", "icosahedron"), ("custom", "custom"), ("trisurfel", "trisurfel"), ("trihexa", "trihexa"), ("sphere", "sphere")) if k in name), "instances")
        tracer = grt.Tracer({"render": {"enable_kernel_timings": True, "primitive_type": prim}})
    g = syn.SimpleGaussians(d12, sph, device=dev)
    g_fd_np, _ = syn.upstream_grads(W, H)
    g_fd = torch.as_tensor(g_fd_np, device=dev)
    if nht:
        g_fd = torch.randn((H, W, 25), device=dev) / (W * H)

    exch = importlib.import_module("3dgrut_amd.dp").GradientExchange(g.parameters(), average=False) if world > 1 else None

    def step():
        g.zero_grad()
        tracer.build_acc(g, rebuild=True)
        out = tracer.render(g, batch, train=True)
        fd = torch.cat([out["pred_features"], out["pred_opacity"]], dim=-1)[0]
        torch.autograd.backward([fd], [g_fd])
        if world > 1:
            exch.reduce()

    for _ in range(args.warmup):
        step()
    tracer.timings
    torch.cuda.synchronize()
    # one instrumented frame (outside the timed region): the traversal's work counters for the byte model
    work = None
    if rank == 0:
        os.environ["GRUT_GRT_COUNT"] = "1"
        step()
        torch.cuda.synchronize()
        del os.environ["GRUT_GRT_COUNT"]
        st = tracer.tracer_wrapper.stats()
        work = {"wave_node_visits": int(st.nodes_visited), "leaf_tests": int(st.candidates), "processed_hits": int(st.processed_hits),
                "list_entries": int(st.list_entries), "packet_tests": int(st.packet_tests), "list_batches": int(st.list_batches),
                "bwd_atomic_instructions": int(st.bwd_atomic_instructions), "bwd_atomic_words": int(st.bwd_atomic_words)}
        step()  # back to the uninstrumented kernels before timing
        tracer.timings
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stages = tracer.timings
    result = None
    if rank == 0:
        P = W * H
        result = {
            "metric": "train rays/sec (3DGRT forward+backward, primary rays; software BVH + per-frame packet lists)", "value": world * P * args.steps / dt,
            "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"3DGRT BVH build + fwd + bwd, {n} Gaussians (cloud B trained-like, seed 42), {W}x{H}, one view per GPU, "
                                   f"SH degree 3, k = 16 hits per trace", "name": name, "parallelism": f"view-dp{world}"},
            "roofline": grt_roofline(work, P, stages), "roofline_backward": grt_backward_roofline(work, P, stages, {"words": work.get("bwd_atomic_words"), "instructions": work.get("bwd_atomic_instructions")} if work else None), "stages_ms": stages, "work": work}
        if emit:
            print(json.dumps(result), flush=True)
    if world > 1 and emit:
        dist.destroy_process_group()
    return result


def hybrid_scene(n, W, H, ms):
    """BASELINE config 5: `n` trained-like Gaussians + a triangle mesh with every primitive type of the playground — a mirror sphere, a
    glass pane, a textured diffuse floor and PBR spheres (metal, transmissive dielectric, fully textured) with a material table,
    textures and an environment map — seen through a 140-degree fisheye from inside the cloud; world-space rays."""
    import workloads.playground_scenes as ps
    syn = importlib.import_module("workloads.synthetic")
    d12, sph = syn.cloud_trained_like(n, seed=42, median_scale=ms)
    Kf = syn.fisheye_intrinsics(W, H, fov_deg=140.0)
    ro, rd = syn.fisheye_rays(W, H, Kf)
    T = syn.orbit_pose(0, n_views=8, radius=2.4).astype(np.float32)   # close to the cloud: the fisheye frame is mostly covered
    ro_w = (ro @ T[:3, :3].T + T[:3, 3]).astype(np.float32).reshape(H, W, 3)
    rd_w = (rd @ T[:3, :3].T).astype(np.float32).reshape(H, W, 3)
    mats = [ps.material(diffuse=(0.9, 0.8, 0.7, 1.0), diffuse_tex=ps.texture(64, 64, 4, 1, 0.2, 1.0)),                        # 0 diffuse floor
            ps.material(diffuse=(0.95, 0.75, 0.3, 1.0), metallic=1.0, roughness=0.25),                                         # 1 metal
            ps.material(diffuse=(0.9, 0.95, 1.0, 1.0), roughness=0.1, transmission=0.85, ior=1.4),                             # 2 transmissive
            ps.material(diffuse=(1.0, 1.0, 1.0, 1.0), emissive=(0.3, 0.25, 0.2), metallic=0.6, roughness=0.8,
                        diffuse_tex=ps.texture(128, 128, 4, 2, 0.1, 1.0), emissive_tex=ps.texture(32, 32, 4, 3, 0.0, 0.5),
                        metallic_roughness_tex=ps.texture(64, 64, 2, 4, 0.05, 1.0), normal_tex=ps.texture(64, 64, 4, 5, 0.3, 1.0))]   # 3 textured
    parts = []
    v, f, nn, uv = ps.uv_sphere((0.0, 0.0, 0.0), 0.45, 24)
    parts.append((v, f, nn, uv, ps.PRIM_MIRROR, 0, 1.0, None))
    for k, (c, m) in enumerate([((0.9, 0.3, 0.2), 1), ((-0.8, 0.5, 0.1), 2), ((0.2, -0.9, 0.3), 3)]):
        v, f, nn, uv = ps.uv_sphere(c, 0.3, 16)
        parts.append((v, f, nn, uv, ps.PRIM_PBR, m, 1.0, None))
    v, f, nn, uv = ps.quad([[-0.7, -0.7, 1.25], [0.7, -0.7, 1.25], [0.7, 0.7, 1.25], [-0.7, 0.7, 1.25]], (0, 0, 1))
    parts.append((v, f, nn, uv, ps.PRIM_GLASS, 0, 1.45, None))
    v, f, nn, uv = ps.quad([[-1.6, -1.6, -1.15], [1.6, -1.6, -1.15], [1.6, 1.6, -1.15], [-1.6, 1.6, -1.15]], (0, 0, 1))
    parts.append((v, f, nn, uv, ps.PRIM_DIFFUSE, 0, 1.0, None))
    return dict(density12=d12, sph=sph, ray_o=np.ascontiguousarray(ro_w), ray_d=np.ascontiguousarray(rd_w), W=W, H=H, mesh=ps._assemble(parts),
                materials=mats, envmap=ps.texture(64, 128, 4, 9, 0.05, 0.9), envmap_offset=np.array([0.13, 0.04], np.float32),
                ray_max_t=np.full((H, W), 1e9, np.float32))


def bench_hybrid(args, dev, n, W, H, ms, emit=True):
    """BASELINE config 5 (forward only, like the reference's playground): world-space fisheye rays through the hybrid tracer."""
    import workloads.playground_scenes as ps
    syn = importlib.import_module("workloads.synthetic")
    pt = importlib.import_module("3dgrut_amd.playground_tracer")
    sc = hybrid_scene(n, W, H, ms)
    mesh = sc["mesh"]
    tr = pt.Tracer({"render": {"enable_kernel_timings": True}})
    g = syn.SimpleGaussians(sc["density12"], sc["sph"], device=dev, requires_grad=False)
    t = lambda a: None if a is None else torch.as_tensor(a, device=dev)
    tr.build_gs_acc(g, rebuild=True)
    tr.build_mesh_acc(t(mesh["vertices"]), t(mesh["triangles"]))
    mats = [dict(diffuse_map=t(x["diffuse_tex"]), emissive_map=t(x["emissive_tex"]), metallic_roughness_map=t(x["metallic_roughness_tex"]),
                 normal_map=t(x["normal_tex"]), diffuse_factor=x["diffuse_factor"], emissive_factor=x["emissive_factor"], metallic_factor=x["metallic_factor"],
                 roughness_factor=x["roughness_factor"], alpha_mode=x["alpha_mode"], alpha_cutoff=x["alpha_cutoff"],
                 transmission_factor=x["transmission_factor"], ior=x["ior"]) for x in sc["materials"]]
    args_t = (g, t(sc["ray_o"])[None], t(sc["ray_d"])[None], ps.OPT_SMOOTH_NORMALS, t(mesh["triangles"]), t(mesh["vertex_normals"]), None, None, t(mesh["prim_type"]))
    kw = dict(material_uv=t(mesh["mat_uv"]), material_id=t(mesh["mat_id"])[:, None], materials=mats, refractive_index=t(mesh["refractive_index"]),
              envmap=t(sc["envmap"]), envmap_offset=t(sc["envmap_offset"]), max_pbr_bounces=7)
    out = None
    for _ in range(args.warmup):
        out = tr.render_playground(*args_t, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        out = tr.render_playground(*args_t, frame_id=k, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    b = out["mirror_bounces"]
    result = {"metric": "rays/sec (hybrid mesh + Gaussian path tracing, forward)", "value": W * H * args.steps / dt, "unit": "rays/s",
              "ms_per_step": dt / args.steps * 1e3, "steps": args.steps, "warmup": args.warmup, "n_gpus": 1, "higher_is_better": True, "dtype": "f32",
              "data": "synthetic",
              "config": {"workload": f"hybrid path tracing, {n} Gaussians + {len(mesh['triangles'])} triangles (mirror sphere, glass pane, textured diffuse floor, "
                                     f"three PBR spheres: metal / transmissive / textured), environment map, fisheye 140 deg, {W}x{H}, smooth normals, "
                                     f"7 PBR bounces, forward only", "name": "c5_hybrid_2m_1080p"},
              "rays_with_mirror_bounce": float((b > 0).float().mean()), "mean_opacity": float(out["pred_opacity"].mean())}
    if emit:
        print(json.dumps(result), flush=True)
    return result


def bench_nht(args, dev, n, W, H, ms, emit=True):
    """3DGUT with neural harmonic features (model.feature_type = nht, defaults of configs/base_gs.yaml: 48 floats per particle, sincos x 1 ->
    24 ray features): forward + backward of one view.  First version of these kernels (one pixel per lane, no checkpoints)."""
    syn = importlib.import_module("workloads.synthetic")
    gt = importlib.import_module("3dgrut_amd.gut_tracer")
    from workloads.scenes import torch_batch
    d12, _ = syn.cloud_trained_like(n, seed=42, median_scale=ms)
    feats = np.random.default_rng(7).uniform(-np.pi / 2, np.pi / 2, size=(n, 48)).astype(np.float32)
    K = syn.pinhole_intrinsics(W, H)
    ro, rd = syn.pinhole_rays(W, H, K)
    batch = torch_batch(dict(rays_ori=ro, rays_dir=rd, T_to_world=syn.orbit_pose(0, n_views=8)[None], intrinsics=K), dev)
    tracer = gt.Tracer({"render": {"enable_kernel_timings": True, "splat": {"k_buffer_size": int(getattr(args, "k_buffer", 0))}},   # (--k-buffer K: the sorted hit buffer in front)
                        "model": {"feature_type": "nht", "nht_features": {"dim": 48, "activation": {"type": "sincos", "num_frequencies": 1},
                                                                          "interpolation_type": "barycentric"}}})
    g = syn.SimpleGaussians(d12, feats, device=dev)
    g_feat = torch.randn((1, H, W, 24), device=dev) / (W * H)
    g_opa = torch.randn((1, H, W, 1), device=dev) / (W * H)

    def step():
        g.zero_grad()
        out = tracer.render(g, batch, train=True)
        torch.autograd.backward([out["pred_features"], out["pred_opacity"]], [g_feat, g_opa])

    for _ in range(args.warmup):
        step()
    tracer.timings
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = tracer.tracer_wrapper.stats()
    result = {"metric": "train rays/sec (3DGUT, neural harmonic features, forward+backward)", "value": W * H * args.steps / dt, "unit": "rays/s",
              "ms_per_step": dt / args.steps * 1e3, "steps": args.steps, "warmup": args.warmup, "n_gpus": 1, "higher_is_better": True, "dtype": "f32",
              "data": "synthetic", "stages_ms": tracer.timings,
              "config": {"workload": f"3DGUT fwd+bwd with neural harmonic features, {n} Gaussians, {W}x{H}, 48 feature floats per particle -> 24 ray features, k_buffer {int(args.k_buffer)}",
                         "name": args.workload},
              "work": {"N": int(st.num_particles), "Nv": int(st.num_visible), "I": int(st.num_intersections)}}
    if emit:
        print(json.dumps(result), flush=True)
    return result


XGMI_LINK_GBS = 153.0        # per link and direction (MI355X_MICROARCH.md); 7 links per GPU, point to point
RING_LINK_EFFICIENCY = 0.65   # assumed share of a link a RCCL ring sustains (no measurement on this pool yet: the driver's SCALE run is the first)


def predicted_exchange(kind, world, n, touched_rows=None):
    """What the exchange step should cost on xGMI: a ring collective is bound by ONE link, so time = bytes a rank forwards / (link rate x
    assumed efficiency).  Printed next to the measured device time so that the first RCCL run can be read against a number."""
    w = max(world, 1)
    rows = n if touched_rows is None else int(touched_rows)
    if kind == "allreduce":                       # five tensors, [N,59] floats
        ring = 2.0 * (w - 1) / w * 236.0 * n
    elif kind == "sharded":                       # reduce-scatter [N,12] + all-to-all factors + all-gather [S,60]
        ring = (w - 1) / w * (48.0 + 12.0 + 240.0) * n
    else:                                         # factored / visible / alllinks: all-reduce [rows,12] + all-gather of [rows+1,3] from every other rank
        fac_bytes = 6.0 if kind == "half" else 12.0   # (half: the view factors travel as scaled IEEE halves)
        ring = 2.0 * (w - 1) / w * 48.0 * rows + (w - 1) * fac_bytes * (rows + 1) + (n if kind == "visible" else 0)
    rate = XGMI_LINK_GBS * RING_LINK_EFFICIENCY
    links = max(1, min(7, w - 1))                 # an MI355X node is a full mesh: one xGMI link to every peer
    # Two bounds for the same bytes: ONE ring (everything a rank forwards crosses one link - the pessimistic model of rounds 3-4) and
    # ALL links (direct pairwise transfers, dp.AllLinksExchange, or RCCL running one ring per link permutation - what it does on a fully
    # connected node): the first RCCL measurement is expected between the two.  The budget for >= 6x at 8 ranks is 0.68 ms exposed.
    return {"ring_bytes_per_rank": int(ring), "assumed_link_GBps": rate, "ms": ring / rate / 1e6, "one_ring_ms": ring / rate / 1e6,
            "all_links_ms": ring / rate / 1e6 / links, "links": links,
            "model": "bytes a rank forwards / (153 GB/s x 0.65) over one ring link (ms, one_ring_ms) or spread over its links to all peers (all_links_ms)"}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this command under torch.distributed.run (one rank per GPU, rendezvous on
    127.0.0.1 at a free port) and pass its output through.  On a box with fewer than N GPUs the RCCL backend cannot place the ranks:
    that is an error unless GRUT_BENCH_BACKEND=gloo asks for the plumbing check (ranks share the devices there are)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("GRUT_BENCH_BACKEND", "nccl") == "nccl":
        raise SystemExit(f"--gpus {n}: only {have} GPU(s) visible (RCCL needs one device per rank; GRUT_BENCH_BACKEND=gloo runs the "
                         f"multi-rank plumbing on the devices there are)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c4_1m_1080p", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short 3DGRT (BASELINE config 3) measurement appended to the default line")
    ap.add_argument("--k-buffer", type=int, default=0, help="3DGUT sorted mode (render.splat.k_buffer_size); 0 = the headline configuration")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started from a bare shell (`python bench.py --gpus N`): launch ourselves, one rank per GPU, and relay rank 0's line
        return self_launch(args.gpus)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks (WORLD_SIZE={world})")
    # one rank per GPU; the modulo only matters for the plumbing check of the multi-rank path on a 1-GPU box
    # (GRUT_BENCH_BACKEND=gloo, two ranks sharing the device), never on a node with >= N GPUs
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("GRUT_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    syn = importlib.import_module("workloads.synthetic")
    gt = importlib.import_module("3dgrut_amd.gut_tracer")
    abi = importlib.import_module("3dgrut_amd._abi")
    from workloads.scenes import torch_batch

    n, W, H, ms = WORKLOADS[args.workload]
    if "grt" in args.workload:
        return bench_grt(args, world, rank, dev, dist, n, W, H, ms)
    if "hybrid" in args.workload:
        return bench_hybrid(args, dev, n, W, H, ms)
    if args.workload.startswith("c4_nht"):
        return bench_nht(args, dev, n, W, H, ms)
    d12, sph = syn.cloud_trained_like(n, seed=42, median_scale=ms)
    K = syn.pinhole_intrinsics(W, H)
    ro, rd = syn.pinhole_rays(W, H, K)
    batch = torch_batch(dict(rays_ori=ro, rays_dir=rd, T_to_world=syn.orbit_pose(rank, n_views=max(world, 8))[None], intrinsics=K), dev)
    splat = {"k_buffer_size": args.k_buffer}
    if os.environ.get("GRUT_BENCH_NO_TILE_CULLING"):   # development aid: the cost of the per-tile culling walks (NOT the headline configuration)
        splat["tile_based_culling"] = False
    tracer = gt.Tracer({"render": {"splat": splat}})
    nat = tracer.tracer_wrapper
    g = syn.SimpleGaussians(d12, sph, device=dev)
    g_fd_np, g_dist_np = syn.upstream_grads(W, H)
    g_fd = torch.as_tensor(g_fd_np, device=dev)
    g_rgb, g_opa = g_fd[None, ..., :3].contiguous(), g_fd[None, ..., 3:].contiguous()
    # the exchange step of the path.  Default: the packed geometric gradient is all-reduced and the SH gradient is
    # rebuilt from gathered per-view factors inside the plugin's backward (dp.FactoredGradientExchange: 168 instead of 413 B per
    # particle over the links at 8 ranks); GRUT_BENCH_EXCHANGE=allreduce selects the plain five-tensor all-reduce for comparison,
    # =visible / =sharded the two variants of 3dgrut_amd/dp.py (tests/test_dp_gloo.py)
    dp = importlib.import_module("3dgrut_amd.dp")
    exch = None
    exchange_kind = "none"
    # ONE-PIECE by default again (round 6): RCCL has never executed this path - no multi-GPU box in six rounds - and the first run should have as
    # few ways to fail as possible: one all-reduce + one all-gather on the compute stream after the backward.  The pipelined form (collectives
    # of particle range i under the finalisation kernels of range i + 1, issued from a callback inside autograd's backward; bit-identical over
    # gloo: tests/test_dp_gloo.py, tests/test_dp_gpu.py) is GRUT_BENCH_EXCHANGE_CHUNKS=4; a hang or a mis-ordering there would not be an exception
    # the net below could catch
    chunks = int(os.environ.get("GRUT_BENCH_EXCHANGE_CHUNKS", "1"))
    auto_exchange = False
    if world > 1:
        exchange_kind = os.environ.get("GRUT_BENCH_EXCHANGE", "auto")
        if exchange_kind == "auto":   # factored, unless the union of the step's views leaves more than a quarter of the scene untouched (decided after the first warm-up step)
            exchange_kind, auto_exchange = "factored", True
        if exchange_kind == "factored":
            tracer.gradient_exchange = dp.FactoredGradientExchange(average=False, timed=True, chunks=chunks)
        elif exchange_kind == "visible":    # the factored exchange on the rows some view touched (OR-reduced masks -> index list)
            tracer.gradient_exchange = dp.VisibleRowsExchange(average=False, timed=True)
        elif exchange_kind == "sharded":    # reduce-scatter + all-to-all + shard-local SH rebuild + all-gather
            tracer.gradient_exchange = dp.ShardedGradientExchange(average=False, timed=True)
        elif exchange_kind == "half":       # view factors as scaled IEEE halves (6 B per particle and view on the links)
            tracer.gradient_exchange = dp.HalfFactorsExchange(average=False, timed=True)
        elif exchange_kind == "alllinks":   # direct pairwise transfers over all 7 xGMI links instead of ring collectives
            tracer.gradient_exchange = dp.AllLinksExchange(average=False, timed=True)
        else:
            exch = dp.GradientExchange(g.parameters(), average=False, timed=True)

    def step():
        g.zero_grad()
        out = tracer.render(g, batch, train=True)
        # upstream gradients per SURVEY §8d: d_rgb, d_opacity ~ N(0,1)/P and no gradient into the hit distance
        # (training never back-props depth: trainer.py:677-748)
        torch.autograd.backward([out["pred_features"], out["pred_opacity"]], [g_rgb, g_opa])
        if exch is not None:  # in-place all-reduce of the Gaussian gradients ([N,59] fp32 in five tensors)
            exch.reduce()

    untouched = None
    exchange_fallback = None
    if world > 1 and exchange_kind == "factored" and chunks > 1:
        # The pipelined exchange has run over gloo only (tests/test_dp_gloo.py, tests/test_dp_gpu.py): no multi-GPU box in five rounds.  Its
        # first step over RCCL runs under a net - a failure is the same exception on every rank (same code, same shapes), so all ranks fall
        # back together to the one-piece exchange, say so in the line, and the scaling curve is still measured.
        try:
            step()
            torch.cuda.synchronize()
        except Exception as e:   # noqa: BLE001
            exchange_fallback = f"pipelined exchange failed on its first step ({type(e).__name__}: {e}); one-piece exchange used"
            if rank == 0:
                print(f"[bench] {exchange_fallback}", file=sys.stderr, flush=True)
            chunks = 1
            tracer.gradient_exchange = dp.FactoredGradientExchange(average=False, timed=True, chunks=1)
    for it in range(args.warmup):
        step()
        if it == 0 and auto_exchange:
            # rows no view of this step touched: OR of the ranks' visibility masks (one byte per particle over the links, once)
            seen = tracer.render(g, batch, train=False)["mog_visibility"].reshape(-1).bool().to(torch.uint8)
            dist.all_reduce(seen, op=dist.ReduceOp.MAX)
            untouched = 1.0 - float(seen.float().mean())
            if untouched > 0.25:
                exchange_kind = "visible"
                tracer.gradient_exchange = dp.VisibleRowsExchange(average=False, timed=True)
    # one instrumented frame outside the timed region: the sweeps count the tile entries they evaluate / accept (roofline.touched_bytes)
    abi.check(nat.lib.gut_profile_enable(nat.handle, 2), "gut_profile_enable")
    step()
    torch.cuda.synchronize()
    work = nat.stats()
    abi.check(nat.lib.gut_profile_enable(nat.handle, 1), "gut_profile_enable")
    step()   # back on the uninstrumented kernels before timing
    abi.check(nat.lib.gut_profile_read(nat.handle, (C.c_float * len(abi.GUT_STAGES))()), "gut_profile_read")   # drop the two frames' stage times
    # Stage breakdown OUTSIDE the timed region: an event pair around every stage leaves ~10 us of idle stream at each of the eight
    # stage boundaries (profiles/r02m_timeline.txt), 3.5 % of the step, which the product's default configuration
    # (enable_kernel_timings off, like the reference's) does not pay.  The timed region below keeps the events of the DOMINANT kernel
    # only: its average launch duration there is what `roofline` prices.
    for _ in range(min(args.steps, 20)):
        step()
    stage_all = (C.c_float * len(abi.GUT_STAGES))()
    abi.check(nat.lib.gut_profile_read(nat.handle, stage_all), "gut_profile_read")
    dom_index = max(range(len(abi.GUT_STAGES)), key=lambda i: stage_all[i])
    abi.check(nat.lib.gut_profile_select(nat.handle, 1 << dom_index), "gut_profile_select")
    step()
    abi.check(nat.lib.gut_profile_read(nat.handle, (C.c_float * len(abi.GUT_STAGES))()), "gut_profile_read")
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    stage_ms = (C.c_float * len(abi.GUT_STAGES))()
    abi.check(nat.lib.gut_profile_read(nat.handle, stage_ms), "gut_profile_read")
    st = nat.stats()
    exchange = None
    if world > 1:   # device time of the exchange step: every rank's average, the slowest reported
        timer = (tracer.gradient_exchange or exch).timer
        ms, payload = timer.collect()
        per_rank = torch.zeros(world, device=dev, dtype=torch.float64)
        per_rank[rank] = ms if ms is not None else -1.0
        dist.all_reduce(per_rank)   # (a gather written as a sum: works on every backend)
        exchange = {"kind": exchange_kind, "ms_per_step_per_rank": [float(x) for x in per_rank.tolist()], "payload_bytes_per_rank": int(payload),
                    "note": "device time between issuing the collectives and their completion on the compute stream (RCCL over xGMI)",
                    "predicted": predicted_exchange(exchange_kind, world, n, getattr(tracer.gradient_exchange, "last_rows", None)),
                    "chunks": chunks if exchange_kind == "factored" else 1, "untouched_fraction": untouched}
        if exchange_fallback:
            exchange["fallback"] = exchange_fallback
    if rank == 0:
        P = W * H
        # every stage from the all-stage pass; the dominant kernel from the timed region itself
        stages = {k: float(stage_all[i]) for i, k in enumerate(abi.GUT_STAGES)}
        stages[abi.GUT_STAGES[dom_index]] = float(stage_ms[dom_index])
        model = byte_model(int(st.num_particles), int(st.num_visible), int(st.num_intersections), P, int(st.key_bits))
        dom = max(stages, key=lambda k: stages[k])
        achieved = model[dom] / (stages[dom] * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(args.workload, {}).get(dom)
            except Exception:
                traffic = None
        touched = touched_bytes(dom, work, P, int(st.num_intersections))
        total_bytes = sum(model.values())
        result = {
            "metric": "train rays/sec (3DGUT forward+backward, primary rays)",
            "value": world * P * args.steps / dt,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"3DGUT fwd+bwd, {n} Gaussians (cloud B trained-like, seed 42), {W}x{H}, one view per GPU, "
                                   f"SH degree 3, k_buffer {args.k_buffer}", "name": args.workload,
                       "parallelism": f"view-dp{world}" + ({"factored": " + RCCL all-reduce [N,12] + all-gather of view factors [N+1,3]",
                                                                  "visible": " + OR-reduced row masks, then RCCL all-reduce / all-gather of the touched rows only",
                                                                  "sharded": " + RCCL reduce-scatter [N,12] + all-to-all of view factors + all-gather of the rebuilt shards",
                                                                  "none": ""}.get(exchange_kind, " + RCCL grad all-reduce [N,59]"))},
            # `achieved` / `frac`: bytes the kernel can actually touch (evaluated entries, counted on the device) / its duration.
            # `model_*`: SURVEY §8d's per-entry byte model applied to ALL I tile entries — an upper bound that charges the 88 % of the
            # entries behind the rays' termination, which no kernel reads.  `traffic`: HBM bytes from rocprofv3 PMC (2 x FETCH_SIZE +
            # WRITE_SIZE, profiles/pmc_traffic.json), `frac_traffic` = traffic / duration / peak.
            "roofline": {"bound": "hbm", "kernel": f"gut_{dom}",
                         "achieved": (touched if touched else model[dom]) / (stages[dom] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (touched if touched else model[dom]) / (stages[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "touched_bytes": touched, "traffic": traffic,
                         "frac_traffic": (traffic / (stages[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "model_bytes": model[dom], "model_frac": achieved / HBM_PEAK_GBS, "kernel_ms": stages[dom],
                         "entries": {"I": int(st.num_intersections), "fwd_evaluated": int(work.fwd_entries_evaluated),
                                     "fwd_accepted": int(work.fwd_entries_accepted), "bwd_evaluated": int(work.bwd_entries_evaluated),
                                     "bwd_accepted": int(work.bwd_entries_accepted)}},
            # the compositing sweeps are bound by fp32 VALU issue, not by HBM: see valu_fraction()
            "valu": valu_fraction(dom, stages[dom]),
            "stages_ms": stages,
            "stages_note": f"{abi.GUT_STAGES[dom_index]}: HIP events inside the timed region; the other stages: an all-stage pass of "
                           f"{min(args.steps, 20)} steps before it (event pairs around every stage cost ~80 us of stream idle per step)",
            "stage_bytes": model,
            "frame_algorithmic_gb": total_bytes / 1e9,
            "frame_hbm_frac": (total_bytes / (dt / args.steps)) / 1e9 / HBM_PEAK_GBS,
            "work": {"N": int(st.num_particles), "Nv": int(st.num_visible), "I": int(st.num_intersections), "P": P,
                     "tiles": int(st.num_tiles), "tile_key_bits": int(st.key_bits)},
        }
        if exchange is not None:
            result["exchange"] = exchange
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline()
        if world == 1 and not args.no_secondary and args.workload == "c4_1m_1080p":
            # the metric's second half.  NeRF-Synthetic Lego is not available offline: a synthetic teacher at BASELINE config 1's scale
            # (100 k Gaussians, 8 views at 400x400) is trained back from a perturbed copy for 500 SelectiveAdam steps through each
            # plugin, the way trainer.py drives it (3dgrut_amd/surrogate.py).  PSNR here is HIP-rendered on both sides;
            # tests/test_optim_gpu.py::test_training_at_config1_scale_recovers_the_teacher certifies the same runs with the oracle
            # (oracle-rendered teacher and result: 18.2 -> 32.6 dB 3DGUT, 16.5 -> 27.9 dB 3DGRT, equal to the HIP numbers to 0.01 dB).
            sur = importlib.import_module("workloads.surrogate")
            result["psnr_surrogate"] = {"data": "synthetic teacher, 100000 Gaussians, 8 views at 400x400, 500 steps (no dataset offline)",
                                        "reference_published": {"3dgut_lego": 36.47, "3dgrt_lego": 36.70, "source": "README.md:362,408 (real data, 30k steps)"}}
            for method in ("3dgut", "3dgrt"):
                t0 = time.time()
                r = sur.train_surrogate(method)
                result["psnr_surrogate"][method] = {"psnr_before_db": r["psnr_hip_before"], "psnr_after_db": r["psnr_hip_after"],
                                                    "wall_s": time.time() - t0}
            # BASELINE config 3 in the same line (the driver runs the default command only): 3DGRT software-BVH primary rays,
            # BVH rebuilt + forward + backward per step, 1 M Gaussians at 800x800 — few steps, ~1 s
            sec = argparse.Namespace(steps=5, warmup=2, workload="c3_grt_1m_800")
            gn, gw, gh, gms = WORKLOADS["c3_grt_1m_800"]
            r = bench_grt(sec, 1, 0, dev, None, gn, gw, gh, gms, name="c3_grt_1m_800", emit=False)
            result["secondary"] = {"c3_grt_1m_800": {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "config", "stages_ms",
                                                                        "roofline", "roofline_backward", "work")}}
            # BASELINE config 2 (1 M Gaussians, 800x800: one wave per half tile leaves the chip short of waves), same code
            # path as the headline, run as its own process so that nothing of this one's state is shared
            # ... and BASELINE config 1 (100 k Gaussians, 400x400), the smallest launch: both run the quarter-tile forward
            import subprocess
            for small in ("c2_1m_800", "c1_100k_400"):
                try:
                    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", small, "--no-cpu-baseline", "--no-secondary",
                                          "--steps", str(args.steps), "--warmup", str(args.warmup)], capture_output=True, text=True, timeout=300).stdout
                    c2 = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
                    result["secondary"][small] = {k: c2[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "config", "stages_ms", "work")}
                except Exception as e:   # the headline line must not depend on it
                    result["secondary"][small] = {"error": repr(e)[:200]}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

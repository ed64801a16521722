#!/usr/bin/env python
"""Generates tests/golden/per_hit_deg{2,4}.npz, adam.npz, camera.npz, projector.npz, grt_proxies.npz, grt_trace.npz and
gut_render.npz from the REFERENCE's own per-hit math, optimizer kernel, camera projection code, projection stage, 3DGRT
proxy kernels and OptiX programs, and 3DGUT projection / render / renderBackward kernels.

Run in the build container only (needs /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

oracle/ref/*.cpp compile threedgrt_tracer/include/3dgrt/kernels/cuda/gaussianParticles.cuh and
threedgut_tracer/include/3dgut/kernels/cuda/models/gaussianParticles.cuh on the host (g++ + a CUDA shim) into
oracle/_ref/*.so; this script feeds them seeded random rays / particles / ray states and stores inputs and outputs.
The reference ships no golden vectors of its own for this path, so these known-answer vectors,
produced by the reference's source, are what pins the oracle (tests/test_oracle_cpu.py).
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")

F = np.float32
MIN_RESPONSE, MIN_ALPHA, MIN_T_GUT, MIN_T_GRT = F(0.0113), F(1.0 / 255.0), F(1e-4), F(1e-3)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def cases(n, seed):
    """Random particles with rays aimed near them so that most cases are accepted hits."""
    r = np.random.default_rng(seed)
    pos = r.uniform(-1, 1, (n, 3))
    scl = np.exp(r.normal(np.log(0.1), 0.6, (n, 3)))
    q = r.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    dens = r.uniform(0.02, 1.0, (n, 1))
    d12 = np.concatenate([pos, dens, q, scl, np.zeros((n, 1))], 1).astype(F)
    ro = r.uniform(-4, 4, (n, 3))
    target = pos + r.normal(size=(n, 3)) * scl.mean(1, keepdims=True) * 1.2
    rd = target - ro
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    sph = (r.normal(size=(n, 48)) * 0.3).astype(F)
    feat = r.uniform(0, 1, (n, 3)).astype(F)
    T = r.uniform(0.05, 1.0, (n, 1))
    state5 = np.concatenate([T, r.uniform(0, 1, (n, 3)) * (1 - T), r.uniform(0, 3, (n, 1)) * (1 - T)], 1).astype(F)
    fin5 = np.concatenate([T * r.uniform(0.0, 0.9, (n, 1)), state5[:, 1:4] + r.uniform(0, 0.5, (n, 3)),
                           state5[:, 4:5] + r.uniform(0, 2, (n, 1))], 1).astype(F)
    grads5 = r.normal(size=(n, 5)).astype(F)
    return dict(ray_o=ro.astype(F), ray_d=rd.astype(F), density12=d12, sph48=sph, feat3=feat, state5=state5, fin5=fin5,
                grads5=grads5)


def run(degree, n=512, seed=1234):
    grt = C.CDLL(os.path.join(REF, f"libref_hit_deg{degree}.so"))
    gut = C.CDLL(os.path.join(REF, f"libref_gut_hit_deg{degree}.so"))
    assert grt.ref_degree() == degree and gut.ref_gut_degree() == degree
    for fn in (grt.ref_particle_response, grt.ref_particle_response_grd):
        fn.restype = C.c_float
    grt.ref_particle_response.argtypes = [C.c_float]
    grt.ref_particle_response_grd.argtypes = [C.c_float] * 3
    c = cases(n, seed + degree)
    out = dict(c)
    out["params"] = np.array([MIN_RESPONSE, MIN_ALPHA, MIN_T_GUT, MIN_T_GRT], F)
    fl = C.c_float
    # ---- 3DGUT per-hit (per-particle radiance) ----
    gut_fwd_state = c["state5"].copy()
    gut_fwd_acc = np.zeros(n, np.int32)
    gut_bwd_state = c["state5"].copy()
    gut_gd = np.zeros((n, 12), F)
    gut_gf = np.zeros((n, 3), F)
    # ---- 3DGRT per-hit (per-ray SH radiance, normals) ----
    grt_state = np.concatenate([c["state5"], np.zeros((n, 3), F)], 1)
    grt_acc = np.zeros(n, np.int32)
    grt_bwd_state = c["state5"].copy()
    grt_gd = np.zeros((n, 12), F)
    grt_gs = np.zeros((n, 48), F)
    sh_out = np.zeros((4, n, 3), F)
    sh_bwd_out = np.zeros((n, 3), F)
    sh_bwd_g = np.zeros((n, 48), F)
    custom_t = np.zeros(n, F)
    custom_ok = np.zeros(n, np.int32)
    inst_t = np.zeros(n, F)
    inst_ok = np.zeros(n, np.int32)
    pray = np.zeros((n, 6), F)
    for i in range(n):
        a = {k: np.ascontiguousarray(v[i]) for k, v in c.items()}
        gut_fwd_acc[i] = gut.ref_gut_process_hit_fwd(_p(a["ray_o"]), _p(a["ray_d"]), _p(a["density12"]), _p(a["feat3"]), fl(MIN_RESPONSE),
                                                     fl(MIN_ALPHA), _p(gut_fwd_state[i]))
        gut.ref_gut_process_hit_bwd(_p(a["ray_o"]), _p(a["ray_d"]), _p(a["density12"]), _p(a["feat3"]), fl(MIN_RESPONSE), fl(MIN_ALPHA),
                                    fl(MIN_T_GUT), _p(gut_bwd_state[i]), _p(a["fin5"]), _p(a["grads5"]), _p(gut_gd[i]), _p(gut_gf[i]))
        grt_acc[i] = grt.ref_process_hit(_p(a["ray_o"]), _p(a["ray_d"]), _p(a["density12"]), _p(a["sph48"]), fl(MIN_RESPONSE), fl(MIN_ALPHA),
                                         3, 1, _p(grt_state[i]))
        grt.ref_process_hit_bwd(_p(a["ray_o"]), _p(a["ray_d"]), _p(a["density12"]), _p(a["sph48"]), fl(MIN_RESPONSE), fl(MIN_ALPHA),
                                fl(MIN_T_GRT), 3, _p(grt_bwd_state[i]), _p(a["fin5"]), _p(a["grads5"]), _p(grt_gd[i]), _p(grt_gs[i]))
        for deg in range(4):
            grt.ref_radiance_from_sph(deg, _p(a["sph48"]), _p(a["ray_d"]), _p(sh_out[deg, i]))
        grt.ref_radiance_from_sph_bwd(3, _p(a["sph48"]), _p(a["ray_d"]), fl(0.37), _p(np.ascontiguousarray(a["grads5"][1:4])),
                                      _p(sh_bwd_out[i]), _p(sh_bwd_g[i]))
        t = fl(0)
        custom_ok[i] = grt.ref_intersect_custom(_p(a["ray_o"]), _p(a["ray_d"]), _p(a["density12"]), fl(0.0), fl(1e6), fl(9.0), C.byref(t))
        custom_t[i] = t.value
        # instance-space ray: o' = (R^T (o - mu)) / (r s), d' = (R^T d) / (r s) with r = 3 (the proxy's kernel scale)
        d12 = c["density12"][i].astype(np.float64)
        w, x, y, z = d12[4:8]
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        po = (R.T @ (c["ray_o"][i] - d12[0:3])) / (3.0 * d12[8:11])
        pd = (R.T @ c["ray_d"][i].astype(np.float64)) / (3.0 * d12[8:11])
        pray[i] = np.concatenate([po, pd]).astype(F)
        inst_ok[i] = grt.ref_intersect_instance(_p(np.ascontiguousarray(pray[i, :3])), _p(np.ascontiguousarray(pray[i, 3:])), fl(0.0), fl(1e6),
                                                fl(1.0), C.byref(t))
        inst_t[i] = t.value
    gray = np.linspace(0, 12, 64).astype(F)
    resp = np.array([grt.ref_particle_response(fl(g)) for g in gray], F)
    resp_grd = np.array([grt.ref_particle_response_grd(fl(g), fl(r), fl(0.7)) for g, r in zip(gray, resp)], F)
    out.update(gut_fwd_state=gut_fwd_state, gut_fwd_accept=gut_fwd_acc, gut_bwd_state=gut_bwd_state, gut_g_density12=gut_gd,
               gut_g_feat3=gut_gf, grt_fwd_state=grt_state, grt_fwd_accept=grt_acc, grt_bwd_state=grt_bwd_state,
               grt_g_density12=grt_gd, grt_g_sph48=grt_gs, sh_radiance=sh_out, sh_bwd_radiance=sh_bwd_out, sh_bwd_g_sph48=sh_bwd_g,
               custom_t=custom_t, custom_ok=custom_ok, instance_ray=pray, instance_t=inst_t, instance_ok=inst_ok,
               gray=gray, response=resp, response_grd=resp_grd)
    path = os.path.join(HERE, f"per_hit_deg{degree}.npz")
    np.savez_compressed(path, **out)
    print(path, f"accepted gut {gut_fwd_acc.mean():.2f} grt {grt_acc.mean():.2f} custom {custom_ok.mean():.2f} instance {inst_ok.mean():.2f}")




def make_adam():
    """tests/golden/adam.npz: three consecutive steps of the reference's selective_adam_update_kernel (oracle/_ref/libref_adam.so)
    on a [N, M] parameter for the row widths of the six Gaussian parameter groups."""
    lib = C.CDLL(os.path.join(REF, "libref_adam.so"))
    r = np.random.default_rng(11)
    out = {}
    for M in (1, 3, 4, 45):
        N = 97
        p = r.normal(size=(N, M)).astype(F)
        m = np.zeros((N, M), F)
        v = np.zeros((N, M), F)
        lr, b1, b2, eps = F(10.0 ** r.uniform(-4, -2)), F(0.9), F(0.999), F(1e-15 if M == 3 else 1e-8)
        out[f"M{M}_hyper"] = np.array([lr, b1, b2, eps], F)
        out[f"M{M}_p0"] = p.copy()
        for step in range(3):
            g = (r.normal(size=(N, M)) * 10.0 ** r.uniform(-6, 0, (N, 1))).astype(F)
            vis = r.uniform(size=N) < 0.6
            lib.ref_selective_adam(_p(p), _p(g), _p(m), _p(v), _p(vis), C.c_float(lr), C.c_float(b1), C.c_float(b2), C.c_float(eps),
                                   C.c_uint32(N), C.c_uint32(M))
            out[f"M{M}_g{step}"], out[f"M{M}_vis{step}"] = g, vis
            out[f"M{M}_p{step + 1}"], out[f"M{M}_m{step + 1}"], out[f"M{M}_v{step + 1}"] = p.copy(), m.copy(), v.copy()
    np.savez_compressed(os.path.join(HERE, "adam.npz"), **out)
    print("wrote adam.npz")


def camera_cases(seed=21, n_points=160):
    """Seeded camera models (three projection models x five shutter types), two sensor poses per case and points scattered
    in and around the view frustum."""
    r = np.random.default_rng(seed)
    cases = []
    W, H = 640, 400
    for model in (0, 1, 2):
        for shutter in range(5):
            prm = np.zeros(33, F)
            prm[0:2] = (W / 2 + r.normal() * 5, H / 2 + r.normal() * 5)
            if model == 0:
                prm[2:4] = (500 + r.normal() * 20, 510 + r.normal() * 20)
                prm[4:10] = r.normal(size=6) * np.array([0.05, 0.02, 0.005, 0.04, 0.015, 0.004])
                prm[10:12] = r.normal(size=2) * 1e-3
                prm[12:16] = r.normal(size=4) * 1e-3
            elif model == 1:
                prm[2:4] = (300 + r.normal() * 10, 305 + r.normal() * 10)
                prm[4:8] = r.normal(size=4) * np.array([0.03, 0.01, 0.003, 0.001])
                prm[16] = 1.3
            else:
                f = 320.0
                bw = np.array([0.0, 1.0 / f, 0.0, 2e-9, 0.0, 1e-15])       # pixel distance -> angle
                fw = np.array([0.0, f, 0.0, -f * f * f * 2e-9 * f, 0.0, 0.0])  # angle -> pixel distance (approximate inverse)
                prm[16] = 1.4
                prm[17] = float(shutter % 2)
                prm[18:24] = bw
                prm[24:30] = fw
                prm[30:33] = (1.0 + r.normal() * 1e-3, r.normal() * 1e-3, r.normal() * 1e-3)
            def pose():
                q = r.normal(size=4) * np.array([0.05, 0.05, 0.05, 0.0]) + np.array([0, 0, 0, 1.0])
                q /= np.linalg.norm(q)
                return np.concatenate([r.normal(size=3) * 0.1, q]).astype(F)
            ps, pe = pose(), pose()
            pts = np.concatenate([r.uniform(-2.5, 2.5, (n_points, 2)), r.uniform(-0.5, 6.0, (n_points, 1))], 1).astype(F)
            cases.append(dict(model=model, shutter=shutter, W=W, H=H, prm=prm, ps=ps, pe=pe, pts=pts))
    return cases


def make_camera():
    """tests/golden/camera.npz: projectPointWithShutter<5> / <0>, sensorPoseInverse, interpolatedSensorPose of the reference
    (oracle/_ref/libref_camera.so = cameraProjections.cuh + sensors.h compiled on the host)."""
    lib = C.CDLL(os.path.join(REF, "libref_camera.so"))
    out = {}
    for k, c in enumerate(camera_cases()):
        n = len(c["pts"])
        for n_iter in (5, 0):
            xy = np.zeros((n, 2), F)
            ok = np.zeros(n, np.int32)
            for i in range(n):
                o = np.zeros(2, F)
                ok[i] = lib.ref_project_point_with_shutter(c["model"], c["shutter"], c["W"], c["H"], _p(c["prm"]), _p(c["ps"]), _p(c["pe"]),
                                                           n_iter, _p(np.ascontiguousarray(c["pts"][i])), C.c_float(0.1), _p(o))
                xy[i] = o
            out[f"c{k}_xy{n_iter}"], out[f"c{k}_ok{n_iter}"] = xy, ok
        inv, mid = np.zeros(7, F), np.zeros(7, F)
        lib.ref_pose_inverse(_p(c["ps"]), _p(inv))
        lib.ref_pose_interpolate(_p(c["ps"]), _p(c["pe"]), C.c_float(0.37), _p(mid))
        out[f"c{k}_inv"], out[f"c{k}_mid"] = inv, mid
    np.savez_compressed(os.path.join(HERE, "camera.npz"), **out)
    print("wrote camera.npz", sum(int(v.sum()) for k, v in out.items() if "_ok" in k), "valid projections")


def projector_cases(seed=33, n=700):
    """Particle clouds in front of the cameras of camera_cases() (one shutter type per model is enough here: the shutter
    logic itself is pinned by camera.npz): anisotropic scales, random rotations, densities on both sides of 1/255."""
    r = np.random.default_rng(seed)
    cases = []
    for c in camera_cases():
        if c["shutter"] not in (4, 0):   # global and rolling top-to-bottom
            continue
        pos = np.concatenate([r.uniform(-2.2, 2.2, (n, 2)), r.uniform(-0.3, 6.0, (n, 1))], 1)
        scl = np.exp(r.normal(np.log(0.05), 0.8, (n, 3)))
        q = r.normal(size=(n, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        dens = np.where(r.uniform(size=(n, 1)) < 0.1, r.uniform(0.0, 0.006, (n, 1)), r.uniform(0.01, 1.0, (n, 1)))
        d12 = np.concatenate([pos, dens, q, scl, np.zeros((n, 1))], 1).astype(F)
        cases.append(dict(c, d12=d12))
    return cases


def make_projector():
    """tests/golden/projector.npz: GUTProjector::eval of the reference (oracle/_ref/libref_projector.so) — tiles count,
    projected centre, conic + opacity, extent, depth and visibility of every particle."""
    lib = C.CDLL(os.path.join(REF, "libref_projector.so"))
    out = {}
    for k, c in enumerate(projector_cases()):
        n = len(c["d12"])
        tc = np.zeros(n, np.uint32); pp = np.zeros((n, 2), F); co = np.zeros((n, 4), F); ex = np.zeros((n, 2), F)
        dp = np.zeros(n, F); vis = np.zeros(n, np.int32)
        lib.ref_project_particles(c["model"], c["shutter"], c["W"], c["H"], _p(c["prm"]), _p(c["ps"]), _p(c["pe"]), C.c_uint32(n), _p(c["d12"]),
                                  _p(tc), _p(pp), _p(co), _p(ex), _p(dp), _p(vis))
        out[f"p{k}_tiles"], out[f"p{k}_pos"], out[f"p{k}_conic"], out[f"p{k}_extent"], out[f"p{k}_depth"], out[f"p{k}_vis"] = tc, pp, co, ex, dp, vis
        # GUTProjector::expand on those outputs, then the stable sort by key of gutRenderer.cu:356-365 (CUB radix sort)
        offsets = np.cumsum(tc, dtype=np.uint64).astype(np.uint32)
        total = int(offsets[-1])
        keys, idx = np.zeros(total, np.uint64), np.zeros(total, np.uint32)
        lib.ref_expand_particles(c["W"], c["H"], C.c_uint32(n), _p(offsets), _p(pp), _p(co), _p(ex), _p(dp), _p(keys), _p(idx))
        order = np.argsort(keys, kind="stable")
        out[f"p{k}_sorted_keys"], out[f"p{k}_sorted_idx"] = keys[order], idx[order]
    np.savez_compressed(os.path.join(HERE, "projector.npz"), **out)
    print("wrote projector.npz:", {k: int((v > 0).sum()) for k, v in out.items() if k.endswith("_tiles")})


def make_grt_proxies():
    """tests/golden/grt_proxies.npz: kernelScale and the enclosing AABB / instance kernels of the reference
    (threedgrt_tracer/src/particlePrimitives.cu through oracle/_ref/libref_grt_proxies.so)."""
    lib = C.CDLL(os.path.join(REF, "libref_grt_proxies.so"))
    lib.ref_kernel_scale.restype = C.c_float
    r = np.random.default_rng(44)
    out = {}
    dens = np.concatenate([r.uniform(0.001, 1.0, 200), [0.0113, 0.0116, 0.5, 1.0]]).astype(F)
    for deg in (0, 1, 2, 3, 4, 5, 8):
        for clamp in (0, 1):
            out[f"ks_deg{deg}_c{clamp}"] = np.array([lib.ref_kernel_scale(C.c_float(d), C.c_float(MIN_RESPONSE), C.c_uint(clamp), C.c_float(deg))
                                                     for d in dens], F)
    out["ks_density"] = dens
    n = 400
    pos = r.uniform(-2, 2, (n, 3)).astype(F)
    q = r.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True); q = q.astype(F)
    scl = np.exp(r.normal(np.log(0.05), 0.8, (n, 3))).astype(F)
    dn = r.uniform(0.005, 1.0, n).astype(F)
    for deg, clamp in ((4, 1), (2, 0)):
        aabb, tr = np.zeros((n, 6), F), np.zeros((n, 12), F)
        lib.ref_enclosing_proxies(C.c_uint(n), _p(pos), _p(q), _p(scl), _p(dn), C.c_float(MIN_RESPONSE), C.c_uint(clamp), C.c_float(deg), _p(aabb), _p(tr))
        out[f"px_deg{deg}_c{clamp}_aabb"], out[f"px_deg{deg}_c{clamp}_transform"] = aabb, tr
    out["px_pos"], out["px_rot"], out["px_scl"], out["px_dns"] = pos, q, scl, dn
    np.savez_compressed(os.path.join(HERE, "grt_proxies.npz"), **out)
    print("wrote grt_proxies.npz")


GRT_TRACE_SCENES = [dict(n=600, width=24, height=16, median_scale=0.10, max_density=0.8, kind="trained", seed=3),
                    dict(n=900, width=20, height=20, median_scale=0.06, max_density=0.99, kind="random", seed=5)]


def grt_trace_upstream(h, w, seed=17):
    r = np.random.default_rng(seed)
    return (r.normal(size=(h, w, 3)).astype(F), r.normal(size=(h, w, 1)).astype(F), (r.normal(size=(h, w, 1)) * 0.1).astype(F))


def make_grt_trace():
    """tests/golden/grt_trace.npz: the reference's 3DGRT forward and backward OptiX programs (raygen round loop, intersection,
    any-hit k-buffer, processHit / processHitBwd) run on the host over an emulated OptiX traversal
    (oracle/ref/ref_grt_trace*.cpp), on the proxy instances of the reference's own instance kernel."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from scenes import make_scene
    px = C.CDLL(os.path.join(REF, "libref_grt_proxies.so"))
    fw = C.CDLL(os.path.join(REF, "libref_grt_trace_deg4.so"))
    bw = C.CDLL(os.path.join(REF, "libref_grt_trace_bwd_deg4.so"))
    out = {}
    for k, kw in enumerate(GRT_TRACE_SCENES):
        sc = make_scene(**kw)
        d12, sph = np.ascontiguousarray(sc["density12"]), np.ascontiguousarray(sc["sph"])
        n, H, W = len(d12), kw["height"], kw["width"]
        pos, rot, scl, dns = (np.ascontiguousarray(d12[:, 0:3]), np.ascontiguousarray(d12[:, 4:8]), np.ascontiguousarray(d12[:, 8:11]),
                              np.ascontiguousarray(d12[:, 3]))
        aabb, tf = np.zeros((n, 6), F), np.zeros((n, 12), F)
        px.ref_enclosing_proxies(C.c_uint(n), _p(pos), _p(rot), _p(scl), _p(dns), C.c_float(MIN_RESPONSE), C.c_uint(1), C.c_float(4), _p(aabb), _p(tf))
        box = np.concatenate([aabb[:, :3].min(0), aabb[:, 3:].max(0)]).astype(F)
        r2w = np.ascontiguousarray(np.asarray(sc["batch"]["T_to_world"][0], F)[:3, :4])
        ro, rd = (np.ascontiguousarray(a.reshape(H, W, 3)) for a in sc["rays"])
        feat, den, hit, nrm = np.zeros((H, W, 3), F), np.zeros((H, W, 1), F), np.zeros((H, W, 2), F), np.zeros((H, W, 3), F)
        cnt, vis = np.zeros((H, W, 1), F), np.zeros(n, np.int32)
        common = (C.c_uint(n), _p(tf), _p(d12), _p(sph), W, H, _p(r2w), _p(ro), _p(rd), _p(box), C.c_float(MIN_T_GRT), C.c_float(MIN_RESPONSE),
                  C.c_float(MIN_ALPHA), C.c_uint(3))
        # the other legal traversal outcome (boxes tested against the ray's shrunk far end): accepted-hit counts only, for the record
        fw.ref_grt_set_box_test_uses_shrunk_tmax(1)
        cnt_shrunk = np.zeros((H, W, 1), F)
        fw.ref_grt_trace_fwd(*common, _p(feat.copy()), _p(den.copy()), _p(hit.copy()), _p(nrm.copy()), _p(cnt_shrunk), _p(vis.copy()))
        out[f"s{k}_hits_count_shrunk_tmax"] = cnt_shrunk
        fw.ref_grt_set_box_test_uses_shrunk_tmax(0)
        fw.ref_grt_trace_fwd(*common, _p(feat), _p(den), _p(hit), _p(nrm), _p(cnt), _p(vis))
        g_rad, g_dns, g_hit = grt_trace_upstream(H, W)
        gd, gs = np.zeros((n, 12), F), np.zeros((n, 48), F)
        bw.ref_grt_trace_bwd(*common, _p(feat), _p(den), _p(hit), _p(g_rad), _p(g_dns), _p(g_hit), _p(gd), _p(gs))
        for name, a in dict(features=feat, density=den, hit_distance=hit, normals=nrm, hits_count=cnt, visibility=vis, grad_density=gd, grad_sph=gs).items():
            out[f"s{k}_{name}"] = a
    np.savez_compressed(os.path.join(HERE, "grt_trace.npz"), **out)
    print("wrote grt_trace.npz; hits per ray:", [float(out[f"s{k}_hits_count"].mean()) for k in range(len(GRT_TRACE_SCENES))])


MESH_PRIMITIVES = {"icosahedron": (1, "IcosaHedron"), "octahedron": (2, "OctraHedron"), "tetrahedron": (3, "TetraHedron"), "diamond": (4, "Diamond"),
                   # checker-only so far (the HIP plugin refuses it): the open two-triangle surfel proxy, traced WITHOUT face culling, with the surfel
                   # branches of processHit / processHitBwd (PipelineParameters::SurfelPrimitive)
                   "trisurfel": (6, "TriSurfel"),
                   # ... and the three back-face-culled rhombi in the proxy's coordinate planes: a ray is offered the SAME particle up to three times
                   "trihexa": (7, "TriHexa")}


def make_grt_trace_mesh():
    """tests/golden/grt_trace_mesh.npz: the reference's 3DGRT forward / backward programs compiled for the TRIANGLE-MESH proxies
    (render.primitive_type icosahedron - the paper's configuration, configs/paper/3dgrt/base_ours_reference.yaml:16 - octahedron,
    tetrahedron, diamond) over the emulated OptiX: the particles' meshes come from the reference's own mesh kernels
    (particlePrimitives.cu:63-496 through oracle/ref/ref_grt_proxies.cpp), the traversal offers every front-facing triangle the ray crosses
    (oracle/ref/ref_grt_emul.inl).  Scene 0 of GRT_TRACE_SCENES for every primitive, scene 1 for the icosahedron too."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from scenes import make_scene
    px = C.CDLL(os.path.join(REF, "libref_grt_proxies.so"))
    px.ref_enclosing_mesh.restype = C.c_uint
    out = {}
    for name, (code, tag) in MESH_PRIMITIVES.items():
        fw = C.CDLL(os.path.join(REF, f"libref_grt_trace_{tag}_deg4.so"))
        bw = C.CDLL(os.path.join(REF, f"libref_grt_trace_bwd_{tag}_deg4.so"))
        for k, kw in enumerate(GRT_TRACE_SCENES[:2 if name == "icosahedron" else 1]):
            sc = make_scene(**kw)
            d12, sph = np.ascontiguousarray(sc["density12"]), np.ascontiguousarray(sc["sph"])
            n, H, W = len(d12), kw["height"], kw["width"]
            pos, rot, scl, dns = (np.ascontiguousarray(d12[:, 0:3]), np.ascontiguousarray(d12[:, 4:8]), np.ascontiguousarray(d12[:, 8:11]),
                                  np.ascontiguousarray(d12[:, 3]))
            verts, tris, nv = np.zeros((n * 12, 3), F), np.zeros((n * 20, 3), np.int32), C.c_uint(0)
            nt = px.ref_enclosing_mesh(code, C.c_uint(n), _p(pos), _p(rot), _p(scl), _p(dns), C.c_float(MIN_RESPONSE), C.c_uint(1), C.c_float(4), _p(verts), _p(tris),
                                       C.byref(nv))
            verts, tris = np.ascontiguousarray(verts[:n * nv.value]), np.ascontiguousarray(tris[:n * nt])
            box = np.concatenate([verts.min(0), verts.max(0)]).astype(F)
            r2w = np.ascontiguousarray(np.asarray(sc["batch"]["T_to_world"][0], F)[:3, :4])
            ro, rd = (np.ascontiguousarray(a.reshape(H, W, 3)) for a in sc["rays"])
            feat, den, hit, nrm = np.zeros((H, W, 3), F), np.zeros((H, W, 1), F), np.zeros((H, W, 2), F), np.zeros((H, W, 3), F)
            cnt, vis = np.zeros((H, W, 1), F), np.zeros(n, np.int32)
            common = (C.c_uint(n), C.c_uint(nt), _p(verts), _p(tris), _p(d12), _p(sph), W, H, _p(r2w), _p(ro), _p(rd), _p(box), C.c_float(MIN_T_GRT),
                      C.c_float(MIN_RESPONSE), C.c_float(MIN_ALPHA), C.c_uint(3))
            fw.ref_grt_trace_fwd_mesh(*common, _p(feat), _p(den), _p(hit), _p(nrm), _p(cnt), _p(vis))
            g_rad, g_dns, g_hit = grt_trace_upstream(H, W)
            gd, gs = np.zeros((n, 12), F), np.zeros((n, 48), F)
            bw.ref_grt_trace_bwd_mesh(*common, _p(feat), _p(den), _p(hit), _p(g_rad), _p(g_dns), _p(g_hit), _p(gd), _p(gs))
            for key, a in dict(features=feat, density=den, hit_distance=hit, hits_count=cnt, visibility=vis, grad_density=gd, grad_sph=gs, scene_box=box).items():
                out[f"{name}_s{k}_{key}"] = a
            print(f"{name} scene {k}: {nt} triangles per particle, hits per ray {cnt.mean():.1f} (max {cnt.max():.0f}), opacity {den.mean():.3f}")
    # render.primitive_type custom: the particles' world boxes (computeGaussianEnclosingAABBKernel) as custom primitives, the world-space
    # intersection program intersectCustomParticle (gaussianParticles.cuh:407-441) - both scenes
    fw = C.CDLL(os.path.join(REF, "libref_grt_trace_Custom_deg4.so"))
    bw = C.CDLL(os.path.join(REF, "libref_grt_trace_bwd_Custom_deg4.so"))
    for k, kw in enumerate(GRT_TRACE_SCENES):
        sc = make_scene(**kw)
        d12, sph = np.ascontiguousarray(sc["density12"]), np.ascontiguousarray(sc["sph"])
        n, H, W = len(d12), kw["height"], kw["width"]
        pos, rot, scl, dns = (np.ascontiguousarray(d12[:, 0:3]), np.ascontiguousarray(d12[:, 4:8]), np.ascontiguousarray(d12[:, 8:11]),
                              np.ascontiguousarray(d12[:, 3]))
        aabb, tf = np.zeros((n, 6), F), np.zeros((n, 12), F)
        px.ref_enclosing_proxies(C.c_uint(n), _p(pos), _p(rot), _p(scl), _p(dns), C.c_float(MIN_RESPONSE), C.c_uint(1), C.c_float(4), _p(aabb), _p(tf))
        box = np.concatenate([aabb[:, :3].min(0), aabb[:, 3:].max(0)]).astype(F)
        r2w = np.ascontiguousarray(np.asarray(sc["batch"]["T_to_world"][0], F)[:3, :4])
        ro, rd = (np.ascontiguousarray(a.reshape(H, W, 3)) for a in sc["rays"])
        feat, den, hit, nrm = np.zeros((H, W, 3), F), np.zeros((H, W, 1), F), np.zeros((H, W, 2), F), np.zeros((H, W, 3), F)
        cnt, vis = np.zeros((H, W, 1), F), np.zeros(n, np.int32)
        common = (C.c_uint(n), _p(aabb), _p(d12), _p(sph), W, H, _p(r2w), _p(ro), _p(rd), _p(box), C.c_float(MIN_T_GRT), C.c_float(MIN_RESPONSE),
                  C.c_float(MIN_ALPHA), C.c_uint(3))
        fw.ref_grt_trace_fwd_custom(*common, _p(feat), _p(den), _p(hit), _p(nrm), _p(cnt), _p(vis))
        g_rad, g_dns, g_hit = grt_trace_upstream(H, W)
        gd, gs = np.zeros((n, 12), F), np.zeros((n, 48), F)
        bw.ref_grt_trace_bwd_custom(*common, _p(feat), _p(den), _p(hit), _p(g_rad), _p(g_dns), _p(g_hit), _p(gd), _p(gs))
        for key, a in dict(features=feat, density=den, hit_distance=hit, hits_count=cnt, visibility=vis, grad_density=gd, grad_sph=gs, scene_box=box, boxes=aabb).items():
            out[f"custom_s{k}_{key}"] = a
        print(f"custom scene {k}: hits per ray {cnt.mean():.1f} (max {cnt.max():.0f}), opacity {den.mean():.3f}")
    np.savez_compressed(os.path.join(HERE, "grt_trace_mesh.npz"), **out)
    print("wrote grt_trace_mesh.npz")


def make_grt_trace_sphere():
    """tests/golden/grt_trace_sphere.npz: the reference's 3DGRT forward / backward programs compiled with PARTICLE_PRIMITIVE_TYPE = MOGTracingSphere
    (render.primitive_type sphere, optixTracer.cpp:189-190) over the emulated OptiX's built-in sphere primitive (oracle/ref/ref_grt_emul.inl: both
    roots of a ray are offered to the any-hit program); centres and radii from the reference's own kernel (computeGaussianEnclosingSphereKernel
    through oracle/ref/ref_grt_proxies.cpp).  Both scenes of GRT_TRACE_SCENES."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from scenes import make_scene
    px = C.CDLL(os.path.join(REF, "libref_grt_proxies.so"))
    fw = C.CDLL(os.path.join(REF, "libref_grt_trace_Sphere_deg4.so"))
    bw = C.CDLL(os.path.join(REF, "libref_grt_trace_bwd_Sphere_deg4.so"))
    out = {}
    for k, kw in enumerate(GRT_TRACE_SCENES):
        sc = make_scene(**kw)
        d12, sph = np.ascontiguousarray(sc["density12"]), np.ascontiguousarray(sc["sph"])
        n, H, W = len(d12), kw["height"], kw["width"]
        pos, rot, scl, dns = (np.ascontiguousarray(d12[:, 0:3]), np.ascontiguousarray(d12[:, 4:8]), np.ascontiguousarray(d12[:, 8:11]),
                              np.ascontiguousarray(d12[:, 3]))
        ctr, rad = np.zeros((n, 3), F), np.zeros(n, F)
        px.ref_enclosing_spheres(C.c_uint(n), _p(pos), _p(rot), _p(scl), _p(dns), C.c_float(MIN_RESPONSE), C.c_uint(1), C.c_float(4), _p(ctr), _p(rad))
        box = np.concatenate([(ctr - rad[:, None]).min(0), (ctr + rad[:, None]).max(0)]).astype(F)
        r2w = np.ascontiguousarray(np.asarray(sc["batch"]["T_to_world"][0], F)[:3, :4])
        ro, rd = (np.ascontiguousarray(a.reshape(H, W, 3)) for a in sc["rays"])
        feat, den, hit, nrm = np.zeros((H, W, 3), F), np.zeros((H, W, 1), F), np.zeros((H, W, 2), F), np.zeros((H, W, 3), F)
        cnt, vis = np.zeros((H, W, 1), F), np.zeros(n, np.int32)
        common = (C.c_uint(n), _p(ctr), _p(rad), _p(d12), _p(sph), W, H, _p(r2w), _p(ro), _p(rd), _p(box), C.c_float(MIN_T_GRT), C.c_float(MIN_RESPONSE),
                  C.c_float(MIN_ALPHA), C.c_uint(3))
        fw.ref_grt_trace_fwd_sphere(*common, _p(feat), _p(den), _p(hit), _p(nrm), _p(cnt), _p(vis))
        g_rad, g_dns, g_hit = grt_trace_upstream(H, W)
        gd, gs = np.zeros((n, 12), F), np.zeros((n, 48), F)
        bw.ref_grt_trace_bwd_sphere(*common, _p(feat), _p(den), _p(hit), _p(g_rad), _p(g_dns), _p(g_hit), _p(gd), _p(gs))
        for key, a in dict(features=feat, density=den, hit_distance=hit, hits_count=cnt, visibility=vis, grad_density=gd, grad_sph=gs, scene_box=box, centers=ctr,
                           radii=rad).items():
            out[f"sphere_s{k}_{key}"] = a
        print(f"sphere scene {k}: hits per ray {cnt.mean():.1f} (max {cnt.max():.0f}), opacity {den.mean():.3f}")
    np.savez_compressed(os.path.join(HERE, "grt_trace_sphere.npz"), **out)
    print("wrote grt_trace_sphere.npz")


def make_grt_trace_bary():
    """tests/golden/grt_trace_bary.npz: the reference's surfel forward pipeline (render.pipeline_type barycentricSurfels:
    barycentricSurfelsOptix.cu - ten hits per trace, the response from the hit triangle's barycentrics) over the emulated OptiX's triangles
    (no face culling), trisurfel meshes and {normal, density} rows from the reference's own kernel.  Both scenes of GRT_TRACE_SCENES.  The
    reference has no backward program for this pipeline."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from scenes import make_scene
    px = C.CDLL(os.path.join(REF, "libref_grt_proxies.so"))
    fw = C.CDLL(os.path.join(REF, "libref_grt_trace_bary_deg4.so"))
    out = {}
    for k, kw in enumerate(GRT_TRACE_SCENES):
        sc = make_scene(**kw)
        d12, sph = np.ascontiguousarray(sc["density12"]), np.ascontiguousarray(sc["sph"])
        n, H, W = len(d12), kw["height"], kw["width"]
        pos, rot, scl, dns = (np.ascontiguousarray(d12[:, 0:3]), np.ascontiguousarray(d12[:, 4:8]), np.ascontiguousarray(d12[:, 8:11]),
                              np.ascontiguousarray(d12[:, 3]))
        verts, tris, nd = np.zeros((n * 4, 3), F), np.zeros((n * 2, 3), np.int32), np.zeros((n, 4), F)
        px.ref_enclosing_trisurfels(C.c_uint(n), _p(pos), _p(rot), _p(scl), _p(dns), C.c_float(MIN_RESPONSE), C.c_uint(1), C.c_float(4), _p(verts), _p(tris), _p(nd))
        box = np.concatenate([verts.min(0), verts.max(0)]).astype(F)
        r2w = np.ascontiguousarray(np.asarray(sc["batch"]["T_to_world"][0], F)[:3, :4])
        ro, rd = (np.ascontiguousarray(a.reshape(H, W, 3)) for a in sc["rays"])
        feat, den, hit, nrm = np.zeros((H, W, 3), F), np.zeros((H, W, 1), F), np.zeros((H, W, 2), F), np.zeros((H, W, 3), F)
        cnt, vis = np.zeros((H, W, 1), F), np.zeros(n, np.int32)
        fw.ref_grt_trace_bary_fwd(C.c_uint(n), _p(verts), _p(tris), _p(nd), _p(d12), _p(sph), W, H, _p(r2w), _p(ro), _p(rd), _p(box), C.c_float(MIN_T_GRT),
                                  C.c_float(MIN_RESPONSE), C.c_float(MIN_ALPHA), C.c_uint(3), _p(feat), _p(den), _p(hit), _p(nrm), _p(cnt), _p(vis))
        for key, a in dict(features=feat, density=den, hit_distance=hit, normals=nrm, hits_count=cnt, visibility=vis, scene_box=box, normal_density=nd).items():
            out[f"bary_s{k}_{key}"] = a
        print(f"barycentricSurfels scene {k}: hits per ray {cnt.mean():.1f} (max {cnt.max():.0f}), opacity {den.mean():.3f}")
    np.savez_compressed(os.path.join(HERE, "grt_trace_bary.npz"), **out)
    print("wrote grt_trace_bary.npz")


def make_grt_trace_nht():
    """tests/golden/grt_trace_nht.npz: the reference's SLANG forward pipeline (referenceSlangOptix.cu: raygen round loop, intersection, any-hit
    k-buffer) in the neural-harmonic-features configuration, on the host over the emulated OptiX (oracle/ref/ref_grt_trace_slang.cpp), on the
    proxy instances of the reference's own instance kernel; the per-hit Slang functions are the stand-in's restatements."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from scenes import make_scene
    px = C.CDLL(os.path.join(REF, "libref_grt_proxies.so"))
    fw = C.CDLL(os.path.join(REF, "libref_grt_trace_slang_deg4.so"))
    assert fw.ref_grt_slang_ray_feature_dim() == 24
    out = {}
    for k, kw in enumerate(GRT_TRACE_SCENES):
        sc = make_scene(**kw)
        d12 = np.ascontiguousarray(sc["density12"])
        n, H, W = len(d12), kw["height"], kw["width"]
        feats = nht_features(n, seed=55 + k)
        pos, rot, scl, dns = (np.ascontiguousarray(d12[:, 0:3]), np.ascontiguousarray(d12[:, 4:8]), np.ascontiguousarray(d12[:, 8:11]),
                              np.ascontiguousarray(d12[:, 3]))
        aabb, tf = np.zeros((n, 6), F), np.zeros((n, 12), F)
        px.ref_enclosing_proxies(C.c_uint(n), _p(pos), _p(rot), _p(scl), _p(dns), C.c_float(MIN_RESPONSE), C.c_uint(1), C.c_float(4), _p(aabb), _p(tf))
        box = np.concatenate([aabb[:, :3].min(0), aabb[:, 3:].max(0)]).astype(F)
        r2w = np.ascontiguousarray(np.asarray(sc["batch"]["T_to_world"][0], F)[:3, :4])
        ro, rd = (np.ascontiguousarray(a.reshape(H, W, 3)) for a in sc["rays"])
        feat, den, hit = np.zeros((H, W, 24), F), np.zeros((H, W, 1), F), np.zeros((H, W, 2), F)
        cnt, vis = np.zeros((H, W, 1), F), np.zeros(n, np.int32)
        fw.ref_grt_trace_slang_fwd(C.c_uint(n), _p(tf), _p(d12), _p(feats), W, H, _p(r2w), _p(ro), _p(rd), _p(box), C.c_float(MIN_T_GRT), C.c_float(MIN_RESPONSE),
                                   C.c_float(MIN_ALPHA), _p(feat), _p(den), _p(hit), _p(cnt), _p(vis))
        for name, a in dict(nht_features=feats, features=feat, density=den, hit_distance=hit, hits_count=cnt, visibility=vis).items():
            out[f"s{k}_{name}"] = a
        print(f"grt nht scene {k}: hits per ray {cnt.mean():.1f}, |features| mean {np.abs(feat).mean():.3f}")
    np.savez_compressed(os.path.join(HERE, "grt_trace_nht.npz"), **out)
    print("wrote grt_trace_nht.npz")


def make_grt_trace_nht_mesh():
    """tests/golden/grt_trace_nht_mesh.npz: the SLANG forward pipeline with neural harmonic features compiled for the ICOSAHEDRON proxies
    (oracle/_ref/libref_grt_trace_slang_IcosaHedron_deg4.so: referenceSlangOptix.cu with PARTICLE_PRIMITIVE_TYPE = MOGTracingIcosaHedron over
    the emulated OptiX's built-in triangles, back faces culled), meshes from the reference's own mesh kernel - model.feature_type nht together
    with render.primitive_type icosahedron (round 5).  Scene 0 of GRT_TRACE_SCENES."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from scenes import make_scene
    px = C.CDLL(os.path.join(REF, "libref_grt_proxies.so"))
    px.ref_enclosing_mesh.restype = C.c_uint
    out = {}
    # round 6: the trihexa proxies (three offers per particle) and OptiX's built-in spheres (two offers) through the same Slang programs
    for prim, tag in (("icosahedron", "IcosaHedron"), ("trihexa", "TriHexa"), ("sphere", "Sphere"), ("custom", "Custom")):
        fw = C.CDLL(os.path.join(REF, f"libref_grt_trace_slang_{tag}_deg4.so"))
        assert fw.ref_grt_slang_ray_feature_dim() == 24
        for k, kw in enumerate(GRT_TRACE_SCENES[:1]):
            sc = make_scene(**kw)
            d12 = np.ascontiguousarray(sc["density12"])
            n, H, W = len(d12), kw["height"], kw["width"]
            feats = nht_features(n, seed=55 + k)
            pos, rot, scl, dns = (np.ascontiguousarray(d12[:, 0:3]), np.ascontiguousarray(d12[:, 4:8]), np.ascontiguousarray(d12[:, 8:11]),
                                  np.ascontiguousarray(d12[:, 3]))
            r2w = np.ascontiguousarray(np.asarray(sc["batch"]["T_to_world"][0], F)[:3, :4])
            ro, rd = (np.ascontiguousarray(a.reshape(H, W, 3)) for a in sc["rays"])
            feat, den, hit = np.zeros((H, W, 24), F), np.zeros((H, W, 1), F), np.zeros((H, W, 2), F)
            cnt, vis = np.zeros((H, W, 1), F), np.zeros(n, np.int32)
            tail = (W, H, _p(r2w), _p(ro), _p(rd))
            tail2 = (C.c_float(MIN_T_GRT), C.c_float(MIN_RESPONSE), C.c_float(MIN_ALPHA), _p(feat), _p(den), _p(hit), _p(cnt), _p(vis))
            if prim == "sphere":
                ctr, rad = np.zeros((n, 3), F), np.zeros(n, F)
                px.ref_enclosing_spheres(C.c_uint(n), _p(pos), _p(rot), _p(scl), _p(dns), C.c_float(MIN_RESPONSE), C.c_uint(1), C.c_float(4), _p(ctr), _p(rad))
                box = np.concatenate([(ctr - rad[:, None]).min(0), (ctr + rad[:, None]).max(0)]).astype(F)
                fw.ref_grt_trace_slang_fwd_sphere(C.c_uint(n), _p(ctr), _p(rad), _p(d12), _p(feats), *tail, _p(box), *tail2)
            elif prim == "custom":   # world boxes + the Slang pipeline's own intersection test (particleDensityHitCustom: restated in the shim)
                aabb, tf = np.zeros((n, 6), F), np.zeros((n, 12), F)
                px.ref_enclosing_proxies(C.c_uint(n), _p(pos), _p(rot), _p(scl), _p(dns), C.c_float(MIN_RESPONSE), C.c_uint(1), C.c_float(4), _p(aabb), _p(tf))
                box = np.concatenate([aabb[:, :3].min(0), aabb[:, 3:].max(0)]).astype(F)
                fw.ref_grt_trace_slang_fwd_custom(C.c_uint(n), _p(aabb), _p(d12), _p(feats), *tail, _p(box), *tail2)
            else:
                code, _ = MESH_PRIMITIVES[prim]
                verts, tris, nv = np.zeros((n * 12, 3), F), np.zeros((n * 20, 3), np.int32), C.c_uint(0)
                nt = px.ref_enclosing_mesh(code, C.c_uint(n), _p(pos), _p(rot), _p(scl), _p(dns), C.c_float(MIN_RESPONSE), C.c_uint(1), C.c_float(4), _p(verts), _p(tris),
                                           C.byref(nv))
                verts, tris = np.ascontiguousarray(verts[:n * nv.value]), np.ascontiguousarray(tris[:n * nt])
                box = np.concatenate([verts.min(0), verts.max(0)]).astype(F)
                fw.ref_grt_trace_slang_fwd_mesh(C.c_uint(n), C.c_uint(nt), _p(verts), _p(tris), _p(d12), _p(feats), *tail, _p(box), *tail2)
            for name, a in dict(nht_features=feats, features=feat, density=den, hit_distance=hit, hits_count=cnt, visibility=vis, scene_box=box).items():
                out[f"{prim}_s{k}_{name}"] = a
            print(f"grt nht {prim} scene {k}: hits per ray {cnt.mean():.1f}, |features| mean {np.abs(feat).mean():.3f}")
    np.savez_compressed(os.path.join(HERE, "grt_trace_nht_mesh.npz"), **out)
    print("wrote grt_trace_nht_mesh.npz")


def make_grt_trace_slang_sh():
    """tests/golden/grt_trace_slang_sh.npz: the reference's SLANG forward pipeline (referenceSlangOptix.cu) with SH radiance — the
    configuration render.pipeline_type = referenceSlang of a model.feature_type = sh run — on the scenes of grt_trace.npz, so that the two
    pipelines' outputs can be laid side by side (they integrate the same function; the plugin serves both names with one set of kernels)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from scenes import make_scene
    px = C.CDLL(os.path.join(REF, "libref_grt_proxies.so"))
    fw = C.CDLL(os.path.join(REF, "libref_grt_trace_slangsh_deg4.so"))
    assert fw.ref_grt_slang_ray_feature_dim() == 3
    ref = np.load(os.path.join(HERE, "grt_trace.npz"))
    out = {}
    for k, kw in enumerate(GRT_TRACE_SCENES):
        sc = make_scene(**kw)
        d12, sph = np.ascontiguousarray(sc["density12"]), np.ascontiguousarray(sc["sph"])
        n, H, W = len(d12), kw["height"], kw["width"]
        pos, rot, scl, dns = (np.ascontiguousarray(d12[:, 0:3]), np.ascontiguousarray(d12[:, 4:8]), np.ascontiguousarray(d12[:, 8:11]),
                              np.ascontiguousarray(d12[:, 3]))
        aabb, tf = np.zeros((n, 6), F), np.zeros((n, 12), F)
        px.ref_enclosing_proxies(C.c_uint(n), _p(pos), _p(rot), _p(scl), _p(dns), C.c_float(MIN_RESPONSE), C.c_uint(1), C.c_float(4), _p(aabb), _p(tf))
        box = np.concatenate([aabb[:, :3].min(0), aabb[:, 3:].max(0)]).astype(F)
        r2w = np.ascontiguousarray(np.asarray(sc["batch"]["T_to_world"][0], F)[:3, :4])
        ro, rd = (np.ascontiguousarray(a.reshape(H, W, 3)) for a in sc["rays"])
        feat, den, hit = np.zeros((H, W, 3), F), np.zeros((H, W, 1), F), np.zeros((H, W, 2), F)
        cnt, vis = np.zeros((H, W, 1), F), np.zeros(n, np.int32)
        fw.ref_grt_trace_slang_fwd(C.c_uint(n), _p(tf), _p(d12), _p(sph), W, H, _p(r2w), _p(ro), _p(rd), _p(box), C.c_float(MIN_T_GRT), C.c_float(MIN_RESPONSE),
                                   C.c_float(MIN_ALPHA), _p(feat), _p(den), _p(hit), _p(cnt), _p(vis))
        for name, a in dict(features=feat, density=den, hit_distance=hit, hits_count=cnt, visibility=vis).items():
            out[f"s{k}_{name}"] = a
        print(f"grt slang-sh scene {k}: hits per ray {cnt.mean():.1f}; against the reference pipeline: max |d features| "
              f"{np.abs(feat - ref[f's{k}_features']).max():.2e}, |d density| {np.abs(den - ref[f's{k}_density']).max():.2e}, "
              f"|d hit distance| {np.abs(hit - ref[f's{k}_hit_distance']).max():.2e}, hit counts differ on "
              f"{int((cnt != ref[f's{k}_hits_count']).sum())} rays")
    np.savez_compressed(os.path.join(HERE, "grt_trace_slang_sh.npz"), **out)
    print("wrote grt_trace_slang_sh.npz")


# ---- the reference's 3DGUT kernels (projection, render, renderBackward) on the host --------------------------------------------
GUT_RENDER_SCENES = [
    dict(n=300, width=40, height=36, median_scale=0.12, max_density=0.7),            # translucent: tens of hits per ray, ragged tiles
    dict(n=1500, width=64, height=48, median_scale=0.06, max_density=0.99, seed=5),  # opaque: rays end on the transmittance threshold
]


# non-pinhole cameras and rolling shutters through the same kernels (tests/scenes.make_camera_scene)
GUT_RENDER_CAMERA_SCENES = [("fisheye", dict(n=1200, w=64, h=48)), ("pinhole_rs", dict(n=1200, w=64, h=48)), ("ftheta", dict(n=1200, w=64, h=48))]


def camera_prm(cam):
    """The 33-float camera parameter block of oracle/ref/ref_camera.cpp from a GrutCamera (inverse of test_oracle_cpu._golden_camera)."""
    prm = np.zeros(33, F)
    prm[0:2], prm[2:4] = list(cam.principal_point), list(cam.focal_length)
    prm[4:10], prm[10:12], prm[12:16] = list(cam.radial), list(cam.tangential), list(cam.thin_prism)
    prm[16], prm[17] = cam.max_angle, float(cam.ftheta_reference_poly)
    prm[18:24], prm[24:30], prm[30:33] = list(cam.ftheta_pixeldist_to_angle), list(cam.ftheta_angle_to_pixeldist), list(cam.ftheta_linear_cde)
    return prm


def gut_render_upstream(h, w, seed=23):
    r = np.random.default_rng(seed)
    return r.normal(size=(h, w, 4)).astype(F), (r.normal(size=(h, w, 1)) * 0.1).astype(F)


def gut_reference_frame(sc, k_buffer=0, backward=True, degree=2, balanced=False):
    """Runs the reference's kernels (oracle/_ref/libref_gut_render_deg2_k{K}.so) on a tests/scenes.make_scene() scene, following the
    launch sequence of GUTRenderer::renderForward / renderBackward (gutRenderer.cu:258-413, 472-505): projectOnTiles, inclusive scan,
    expandTileProjections, stable sort by key, tile ranges, render, renderBackward."""
    lib = C.CDLL(os.path.join(REF, f"libref_gut_render_deg{degree}_k{k_buffer}.so"))
    plib = C.CDLL(os.path.join(REF, "libref_projector.so"))   # expandTileProjections (touches no particle data)
    assert lib.ref_gut_k_buffer_size() == k_buffer and lib.ref_gut_kernel_degree() == degree
    W, H = sc["W"], sc["H"]
    d12, sph = np.ascontiguousarray(sc["density12"], F), np.ascontiguousarray(sc["sph"], F)
    n = len(d12)
    cam = sc["cam"]
    prm = camera_prm(cam)
    ps, pe = np.asarray(sc["pose_start"], F), np.asarray(sc["pose_end"], F)
    ro, rd = (np.ascontiguousarray(a, F).reshape(H, W, 3) for a in sc["rays"])
    o = dict(tiles_count=np.zeros(n, np.uint32), proj_pos=np.zeros((n, 2), F), conic_opacity=np.zeros((n, 4), F), extent=np.zeros((n, 2), F),
             depth=np.zeros(n, F), features=np.zeros((n, 3), F), visibility=np.zeros(n, np.int32))
    lib.ref_gut_project(int(cam.model), int(cam.shutter), W, H, _p(prm), _p(ps), _p(pe), C.c_uint32(n), _p(d12), _p(sph), 3, _p(o["tiles_count"]), _p(o["proj_pos"]),
                        _p(o["conic_opacity"]), _p(o["extent"]), _p(o["depth"]), _p(o["features"]), _p(o["visibility"]))
    offsets = np.cumsum(o["tiles_count"], dtype=np.uint64).astype(np.uint32)
    total = int(offsets[-1])
    keys, idx = np.zeros(total, np.uint64), np.zeros(total, np.uint32)
    plib.ref_expand_particles(W, H, C.c_uint32(n), _p(offsets), _p(o["proj_pos"]), _p(o["conic_opacity"]), _p(o["extent"]), _p(o["depth"]),
                              _p(keys), _p(idx))
    order = np.argsort(keys, kind="stable")                      # cub::DeviceRadixSort::SortPairs, gutRenderer.cu:356-365
    o["sorted_idx"] = np.ascontiguousarray(idx[order])
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    tile_of = (keys[order] >> np.uint64(32)).astype(np.int64)     # computeSortedTileRangeIndices, gutRenderer.cu:368-373
    o["tile_ranges"] = np.stack([np.searchsorted(tile_of, np.arange(tiles), "left"),
                                 np.searchsorted(tile_of, np.arange(tiles), "right")], 1).astype(np.uint32)
    # tiles without entries are not written by the kernel and keep the zeros of the freshly sized buffer (gutRenderer.cu:160-161)
    o["tile_ranges"][o["tile_ranges"][:, 0] == o["tile_ranges"][:, 1]] = 0
    lo, hi = np.full(3, -1e6, F), np.full(3, 1e6, F)             # the scene box SplatRaster::trace passes, splatRaster.cpp:240
    o["feat_density"], o["hit_distance"], o["hit_count"] = np.zeros((H, W, 4), F), np.full((H, W, 1), 1e6, F), np.zeros((H, W, 1), F)
    common = (W, H, _p(ps), _p(pe), _p(lo), _p(hi), C.c_uint32(n), _p(d12), _p(sph), 3, _p(o["tile_ranges"]), _p(o["sorted_idx"]),
              _p(o["features"]), _p(ro), _p(rd))
    lib.ref_gut_render_fwd(*common, _p(o["feat_density"]), _p(o["hit_distance"]), _p(o["hit_count"]))
    if balanced:   # the reference's other forward kernel (render.splat.fine_grained_load_balancing), K = 0 builds only
        o["balanced_feat_density"], o["balanced_hit_distance"] = np.zeros((H, W, 4), F), np.full((H, W, 1), 1e6, F)
        o["balanced_hit_count"] = np.zeros((H, W, 1), F)
        lib.ref_gut_render_fwd_balanced(*common, _p(o["balanced_feat_density"]), _p(o["balanced_hit_distance"]), _p(o["balanced_hit_count"]))
    if backward:
        assert k_buffer == 0, "the K > 0 backward is Slang autodiff output (not in the checkout)"
        gfd, gdist = gut_render_upstream(H, W)
        o["grad_density"], o["grad_features"] = np.zeros((n, 12), F), np.zeros((n, 3), F)
        gsph = np.zeros_like(sph)     # untouched by renderBackward in the SH configuration (written by projectBackward)
        lib.ref_gut_render_bwd(*common, _p(o["feat_density"]), _p(gfd), _p(o["hit_distance"]), _p(gdist), _p(o["grad_density"]), _p(gsph),
                               _p(o["grad_features"]))
        assert not gsph.any()
    return o


def gut_standin_check(lib, n=4000, seed=9):
    """Largest difference between the Slang stand-in (oracle/ref/shim/threedgutSlang.cuh) and the reference's CUDA twin of the same
    per-hit math, on random (ray, particle) pairs about half of which are accepted."""
    r = np.random.default_rng(seed)
    pos = r.uniform(-1, 1, (n, 3)); scl = np.exp(r.normal(np.log(0.25), 0.6, (n, 3)))
    q = r.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    d12 = np.concatenate([pos, r.uniform(0.002, 1.0, (n, 1)), q, scl, np.zeros((n, 1))], 1).astype(F)
    ro = r.uniform(-1, 1, (n, 3)) + np.array([0, 0, -4.0])
    rd = pos + r.normal(size=(n, 3)) * scl * 1.8 - ro
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    ro, rd, feat = ro.astype(F), rd.astype(F), r.uniform(-0.3, 1.2, (n, 3)).astype(F)
    a, b = C.c_uint32(0), C.c_uint32(0)
    lib.ref_gut_standin_max_error.restype = C.c_float
    err = lib.ref_gut_standin_max_error(C.c_uint32(n), _p(ro), _p(rd), _p(d12), _p(feat), C.byref(a), C.byref(b))
    return float(err), a.value, b.value


def make_gut_render():
    """tests/golden/gut_render.npz: the reference's projectOnTiles / render / renderBackward kernels, with the real particle class,
    tile loop, hit k-buffer and ray payload code under them, run on the host (oracle/ref/ref_gut_render.cpp) — every binning
    product, the rendered outputs for K = 0 and K = 16, and the K = 0 gradients renderBackward accumulates; then K = 0 frames
    through the fisheye, distorted rolling-shutter pinhole and f-theta cameras.  (One field is not reproducible between runs
    of this script: the visibility flag of a particle whose unscented projection failed is computed by the reference from an
    uninitialised covariance, gutProjector.cuh:246-275; the tests only use it where it is defined.)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from scenes import make_scene
    err, acc_standin, acc_twin = gut_standin_check(C.CDLL(os.path.join(REF, "libref_gut_render_deg2_k0.so")))
    assert acc_standin == acc_twin and 1000 < acc_twin < 3500 and err < 2e-6, (err, acc_standin, acc_twin)
    out = dict(standin_check=np.array([err, acc_standin, acc_twin], np.float64))
    for k, kw in enumerate(GUT_RENDER_SCENES):
        sc = make_scene(**kw)
        o = gut_reference_frame(sc, 0, balanced=True)
        for name, a in o.items():
            out[f"s{k}_{name}"] = a
        print(f"scene {k}: |renderBalanced - render| rgb/opacity {np.abs(o['balanced_feat_density'] - o['feat_density']).max():.3g}, "
              f"distance {np.abs(o['balanced_hit_distance'] - o['hit_distance']).max():.3g}, "
              f"hit counts differ on {int((o['balanced_hit_count'] != o['hit_count']).sum())} pixels")
        o16 = gut_reference_frame(sc, 16, backward=False)
        for name in ("feat_density", "hit_distance", "hit_count"):
            out[f"s{k}_k16_{name}"] = o16[name]
        for kk in (4, 8):   # the other two buffer sizes the plugin dispatches
            ok = gut_reference_frame(sc, kk, backward=False)
            for name in ("feat_density", "hit_distance", "hit_count"):
                out[f"s{k}_k{kk}_{name}"] = ok[name]
        print(f"scene {k}: {int(o['tiles_count'].sum())} tile entries, opacity {o['feat_density'][..., 3].mean():.3f}, "
              f"hits/ray {o['hit_count'].mean():.1f}, |K16 - K0| {np.abs(o16['feat_density'] - o['feat_density']).max():.3g}")
    # the quartic kernel (particle_kernel_degree 4) on the first scene; its stand-in is cross-checked like the quadratic one
    err4, a4, b4 = gut_standin_check(C.CDLL(os.path.join(REF, "libref_gut_render_deg4_k0.so")))
    assert a4 == b4 and 1000 < b4 < 3500 and err4 < 2e-6, (err4, a4, b4)
    out["standin_check_deg4"] = np.array([err4, a4, b4], np.float64)
    o = gut_reference_frame(make_scene(**GUT_RENDER_SCENES[0]), 0, degree=4)
    for name, a in o.items():
        out[f"deg4_{name}"] = a
    print(f"degree 4: {int(o['tiles_count'].sum())} tile entries, opacity {o['feat_density'][..., 3].mean():.3f}, stand-in {err4:.2g}")
    from scenes import make_camera_scene
    for kind, kw in GUT_RENDER_CAMERA_SCENES:
        sc = make_camera_scene(kind, **kw)
        sc["W"], sc["H"] = kw["w"], kw["h"]
        o = gut_reference_frame(sc, 0)
        for name, a in o.items():
            out[f"{kind}_{name}"] = a
        print(f"camera scene {kind}: {int(o['tiles_count'].sum())} tile entries, opacity {o['feat_density'][..., 3].mean():.3f}")
    np.savez_compressed(os.path.join(HERE, "gut_render.npz"), **out)
    print("wrote gut_render.npz; stand-in vs CUDA twin:", err, acc_standin, acc_twin)


# ---- the reference's hybrid mesh + Gaussian path tracer (playground) on the host ----------------------------------------------------
class RefMaterial(C.Structure):   # oracle/ref/ref_playground.cpp: RefMaterial
    _fields_ = [("diffuse_tex", C.c_void_p), ("emissive_tex", C.c_void_p), ("mr_tex", C.c_void_p), ("normal_tex", C.c_void_p),
                ("diffuse_hw", C.c_int32 * 2), ("emissive_hw", C.c_int32 * 2), ("mr_hw", C.c_int32 * 2), ("normal_hw", C.c_int32 * 2),
                ("diffuse_factor", C.c_float * 4), ("emissive_factor", C.c_float * 3), ("metallic", C.c_float), ("roughness", C.c_float),
                ("transmission", C.c_float), ("ior", C.c_float), ("alpha_cutoff", C.c_float), ("alpha_mode", C.c_uint32)]


def nht_features(n, seed=77, dim=48):
    """[n, dim] per-particle neural-harmonic features in the range the model initialises them to (configs/base_gs.yaml:97-99)."""
    return np.random.default_rng(seed).uniform(-np.pi / 2, np.pi / 2, size=(n, dim)).astype(F)


def gut_reference_frame_nht(sc, feats, k_buffer=0):
    """The reference's 3DGUT kernels built for model.feature_type = nht (oracle/_ref/libref_gut_render_nht_deg2_k0.so: FEATURE_TRANSFORM_TYPE 1,
    48 floats per particle = 4 tetrahedron vertices x 12, sincos x 1 frequency -> 24 ray features): projectOnTiles (no per-particle
    radiance), expansion, sort, ranges, render -> [H,W,25] features + opacity, hit distance, hit count."""
    lib = C.CDLL(os.path.join(REF, f"libref_gut_render_nht_deg2_k{k_buffer}.so"))
    plib = C.CDLL(os.path.join(REF, "libref_projector.so"))
    nf = lib.ref_gut_ray_feature_dim()
    assert nf == 24 and lib.ref_gut_particle_feature_dim() == feats.shape[1] == 48 and lib.ref_gut_k_buffer_size() == k_buffer
    W, H = sc["W"], sc["H"]
    d12, feats = np.ascontiguousarray(sc["density12"], F), np.ascontiguousarray(feats, F)
    n = len(d12)
    cam = sc["cam"]
    prm = camera_prm(cam)
    ps, pe = np.asarray(sc["pose_start"], F), np.asarray(sc["pose_end"], F)
    ro, rd = (np.ascontiguousarray(a, F).reshape(H, W, 3) for a in sc["rays"])
    o = dict(tiles_count=np.zeros(n, np.uint32), proj_pos=np.zeros((n, 2), F), conic_opacity=np.zeros((n, 4), F), extent=np.zeros((n, 2), F),
             depth=np.zeros(n, F), visibility=np.zeros(n, np.int32))
    unused = np.zeros((n, nf), F)   # the per-particle feature cache of the SH configuration: not written with per-ray features (gutProjector.cuh:306)
    lib.ref_gut_project(int(cam.model), int(cam.shutter), W, H, _p(prm), _p(ps), _p(pe), C.c_uint32(n), _p(d12), _p(feats), 3, _p(o["tiles_count"]), _p(o["proj_pos"]),
                        _p(o["conic_opacity"]), _p(o["extent"]), _p(o["depth"]), _p(unused), _p(o["visibility"]))
    assert not unused.any()
    offsets = np.cumsum(o["tiles_count"], dtype=np.uint64).astype(np.uint32)
    total = int(offsets[-1])
    keys, idx = np.zeros(total, np.uint64), np.zeros(total, np.uint32)
    plib.ref_expand_particles(W, H, C.c_uint32(n), _p(offsets), _p(o["proj_pos"]), _p(o["conic_opacity"]), _p(o["extent"]), _p(o["depth"]),
                              _p(keys), _p(idx))
    order = np.argsort(keys, kind="stable")
    o["sorted_idx"] = np.ascontiguousarray(idx[order])
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    tile_of = (keys[order] >> np.uint64(32)).astype(np.int64)
    o["tile_ranges"] = np.stack([np.searchsorted(tile_of, np.arange(tiles), "left"),
                                 np.searchsorted(tile_of, np.arange(tiles), "right")], 1).astype(np.uint32)
    o["tile_ranges"][o["tile_ranges"][:, 0] == o["tile_ranges"][:, 1]] = 0
    lo, hi = np.full(3, -1e6, F), np.full(3, 1e6, F)
    o["feat_density"], o["hit_distance"], o["hit_count"] = np.zeros((H, W, nf + 1), F), np.full((H, W, 1), 1e6, F), np.zeros((H, W, 1), F)
    lib.ref_gut_render_fwd(W, H, _p(ps), _p(pe), _p(lo), _p(hi), C.c_uint32(n), _p(d12), _p(feats), 3, _p(o["tile_ranges"]), _p(o["sorted_idx"]),
                           None, _p(ro), _p(rd), _p(o["feat_density"]), _p(o["hit_distance"]), _p(o["hit_count"]))
    return o


def make_gut_nht():
    """tests/golden/gut_nht.npz: the reference's 3DGUT forward in its neural-harmonic-features configuration (see gut_reference_frame_nht;
    the feature model itself — barycentric interpolation in the canonical tetrahedron + sincos — is the stand-in's restatement of
    neuralHarmonicFeaturesParticle.slang, the renderer around it is the reference's code)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from scenes import make_scene
    out = {}
    for k, kw in enumerate(GUT_RENDER_SCENES):
        sc = make_scene(**kw)
        feats = nht_features(len(sc["density12"]), seed=77 + k)
        o = gut_reference_frame_nht(sc, feats)
        out[f"s{k}_features"] = feats
        for name in ("tiles_count", "sorted_idx", "tile_ranges", "feat_density", "hit_distance", "hit_count"):
            out[f"s{k}_{name}"] = o[name]
        print(f"nht scene {k}: opacity {o['feat_density'][..., -1].mean():.3f}, |features| mean {np.abs(o['feat_density'][..., :-1]).mean():.3f}, "
              f"hits/ray {o['hit_count'].mean():.1f}")
        # round 6: the same frames through the sorted hit buffer (GAUSSIAN_K_BUFFER_SIZE 4 / 16 with the feature macros)
        for K in (4, 16):
            ok = gut_reference_frame_nht(sc, feats, k_buffer=K)
            assert np.array_equal(ok["sorted_idx"], o["sorted_idx"])
            for name in ("feat_density", "hit_distance", "hit_count"):
                out[f"s{k}_k{K}_{name}"] = ok[name]
            print(f"nht scene {k} K={K}: max |features - unsorted| {np.abs(ok['feat_density'] - o['feat_density']).max():.3f}")
    np.savez_compressed(os.path.join(HERE, "gut_nht.npz"), **out)
    print("wrote gut_nht.npz")


def playground_reference(sc, opts, bounces, frame):
    """One frame of a tests/playground_scenes.make_playground_scene() scene through the reference's own programs
    (oracle/_ref/libref_playground_deg4.so) over the proxy instances of the reference's instance kernel."""
    px = C.CDLL(os.path.join(REF, "libref_grt_proxies.so"))
    pg = C.CDLL(os.path.join(REF, "libref_playground_deg4.so"))
    d12, sph = sc["density12"], sc["sph"]
    n = len(d12)
    pos, rot, scl, dns = (np.ascontiguousarray(d12[:, 0:3]), np.ascontiguousarray(d12[:, 4:8]), np.ascontiguousarray(d12[:, 8:11]), np.ascontiguousarray(d12[:, 3]))
    aabb, tf = np.zeros((n, 6), F), np.zeros((n, 12), F)
    px.ref_enclosing_proxies(C.c_uint(n), _p(pos), _p(rot), _p(scl), _p(dns), C.c_float(MIN_RESPONSE), C.c_uint(1), C.c_float(4), _p(aabb), _p(tf))
    box = np.concatenate([aabb[:, :3].min(0), aabb[:, 3:].max(0)]).astype(F)
    H, W, m = sc["H"], sc["W"], sc["mesh"]
    ro, rd = sc["ray_o"].copy(), sc["ray_d"].copy()   # overwritten with the last traced segment, like the reference's buffers
    rgb, alpha = np.zeros((H, W, 3), F), np.zeros((H, W, 1), F)
    mats = (RefMaterial * max(len(sc["materials"]), 1))()
    for i, mt in enumerate(sc["materials"]):
        r = mats[i]
        for name, key in (("diffuse", "diffuse_tex"), ("emissive", "emissive_tex"), ("mr", "metallic_roughness_tex"), ("normal", "normal_tex")):
            t = mt[key]
            setattr(r, name + "_tex", t.ctypes.data if t is not None else None)
            hw = getattr(r, name + "_hw")
            hw[0], hw[1] = (t.shape[0], t.shape[1]) if t is not None else (0, 0)
        for k in range(4):
            r.diffuse_factor[k] = float(mt["diffuse_factor"][k])
        for k in range(3):
            r.emissive_factor[k] = float(mt["emissive_factor"][k])
        r.metallic, r.roughness, r.transmission, r.ior = mt["metallic_factor"], mt["roughness_factor"], mt["transmission_factor"], mt["ior"]
        r.alpha_cutoff, r.alpha_mode = mt["alpha_cutoff"], mt["alpha_mode"]
    env = sc["envmap"]
    pg.ref_playground_trace(C.c_uint(n), _p(tf), _p(d12), _p(sph), W, H, _p(ro), _p(rd), _p(sc["ray_max_t"]), _p(box), C.c_float(MIN_T_GRT), C.c_uint(3),
                            C.c_uint(frame), C.c_uint(len(m["vertices"])), _p(m["vertices"]), C.c_uint(len(m["triangles"])), _p(m["triangles"]),
                            _p(m["vertex_normals"]), _p(m["vertex_tangents"]), _p(m["vertex_has_tangents"]), _p(m["prim_type"]), _p(m["mat_uv"]),
                            _p(m["mat_id"]), _p(m["refractive_index"]), C.c_uint(len(sc["materials"])), mats, _p(env), env.shape[0], env.shape[1],
                            _p(sc["envmap_offset"]), C.c_uint(opts), C.c_uint(bounces), _p(rgb), _p(alpha))
    return dict(rgb=rgb, alpha=alpha, last_o=ro, last_d=rd)


def playground_standin_check(n=3000, seed=12):
    """The Slang stand-in of the playground build (shim/3dgrt_slang/...) against the reference's hand-written CUDA twin (processHit of
    3dgrt/kernels/cuda/gaussianParticles.cuh through libref_hit_deg4.so): same rays, same particles -> (largest state difference, accepted by
    the stand-in, accepted by the twin)."""
    pg = C.CDLL(os.path.join(REF, "libref_playground_deg4.so"))
    grt = C.CDLL(os.path.join(REF, "libref_hit_deg4.so"))
    pg.ref_playground_standin_process_hit.restype = C.c_float
    c = cases(n, seed)
    worst, acc_s, acc_t = 0.0, 0, 0
    for i in range(n):
        o, d, prm, sph = (np.ascontiguousarray(c[k][i]) for k in ("ray_o", "ray_d", "density12", "sph48"))
        T, D = np.ones(1, F), np.zeros(1, F)
        w = pg.ref_playground_standin_process_hit(_p(o), _p(d), _p(prm), _p(T), _p(D))
        rad = np.zeros(3, F)
        pg.ref_playground_standin_integrate(_p(d), C.c_float(w), _p(sph), C.c_uint(3), _p(rad))
        st = np.array([1, 0, 0, 0, 0, 0, 0, 0], F)   # transmittance, radiance(3), depth, normal(3)
        acc = grt.ref_process_hit(_p(o), _p(d), _p(prm), _p(sph), C.c_float(MIN_RESPONSE), C.c_float(MIN_ALPHA), C.c_int(3), C.c_int(0), _p(st))
        acc_s += int(w > 0)
        acc_t += int(acc)
        worst = max(worst, abs(float(T[0]) - float(st[0])), float(np.abs(rad - st[1:4]).max()), abs(float(D[0]) - float(st[4])))
    return worst, acc_s, acc_t


def make_playground():
    """tests/golden/playground.npz: the reference's hybrid mesh + Gaussian path tracer (playgroundKernel.cu: raygen path loop, closest-hit
    material dispatch, miss; materials.cuh; trace.cuh; 3dgrtTracer.cuh) run on the host over an emulated OptiX (oracle/ref/ref_playground.cpp)
    on the seeded scenes of tests/playground_scenes.py: mirror / glass / textured diffuse, five PBR materials (metal, transmissive dielectric,
    fully textured with tangents, alpha-masked, alpha-blended), smooth and hard normals, textures on and off, with and without Gaussians."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import playground_scenes as ps
    out = {}
    for name, kind, opts, bounces, frame in ps.GOLDEN_CASES:
        sc = ps.make_playground_scene(kind)
        o = playground_reference(sc, opts, bounces, frame)
        for k, a in o.items():
            out[f"{name}_{k}"] = a
    worst, acc_s, acc_t = playground_standin_check()
    out["standin_vs_cuda_twin"] = np.array([worst, acc_s, acc_t], np.float64)
    np.savez_compressed(os.path.join(HERE, "playground.npz"), **out)
    print("wrote playground.npz; Slang stand-in vs CUDA twin: largest difference", worst, "accepted", acc_s, acc_t)


if __name__ == "__main__":
    import sys
    only = [a for a in sys.argv[1:] if a.startswith("--only=")]
    which = only[0][len("--only="):].split(",") if only else ["per_hit", "adam", "camera", "projector", "grt_proxies", "grt_trace", "grt_trace_mesh", "gut_render", "playground", "gut_nht", "grt_trace_nht", "grt_trace_nht_mesh", "grt_trace_slang_sh", "grt_trace_sphere", "grt_trace_bary"]
    if "--adam-only" in sys.argv:
        which = ["adam"]
    if "per_hit" in which:
        for d in (2, 4):
            run(d)
    if "adam" in which:
        make_adam()
    if "camera" in which:
        make_camera()
    if "projector" in which:
        make_projector()
    if "grt_proxies" in which:
        make_grt_proxies()
    if "grt_trace" in which:
        make_grt_trace()
    if "grt_trace_mesh" in which:
        make_grt_trace_mesh()
    if "gut_render" in which:
        make_gut_render()
    if "playground" in which:
        make_playground()
    if "gut_nht" in which:
        make_gut_nht()
    if "grt_trace_nht" in which:
        make_grt_trace_nht()
    if "grt_trace_nht_mesh" in which:
        make_grt_trace_nht_mesh()
    if "grt_trace_slang_sh" in which:
        make_grt_trace_slang_sh()
    if "grt_trace_sphere" in which:
        make_grt_trace_sphere()
    if "grt_trace_bary" in which:
        make_grt_trace_bary()

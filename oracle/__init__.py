"""CPU oracle (TEST INFRASTRUCTURE ONLY).

numpy-facing wrapper of oracle/gut_oracle.c and oracle/grt_oracle.c, the CPU restatement of the
reference's 3DGUT / 3DGRT algorithms.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package — the product package (3dgrut_amd) never does, and has
no CPU fallback.

Two flavours are built by `make -C oracle`: liboracle32.so (float, mirrors the arithmetic type of
the reference) and liboracle64.so (double, for finite-difference gradient checks).
"""
from __future__ import annotations

import ctypes as C
import importlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_abi = importlib.import_module("3dgrut_amd._abi")  # struct mirrors of include/grut_amd.h only
GutConfig, GrutCamera, GrtConfig = _abi.GutConfig, _abi.GrutCamera, _abi.GrtConfig

_libs = {}


def build(force: bool = False):
    """Compile the oracle with gcc (seconds)."""
    out = os.path.join(_HERE, "_build", "liboracle32.so")
    if force or not os.path.exists(out) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(out)
            for f in ("gut_oracle.c", "grt_oracle.c", "orc_math.h")):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"] if force else ["make", "-C", _HERE, "-s"])


def lib(dtype=np.float32):
    key = np.dtype(dtype).itemsize
    if key not in _libs:
        build()
        name = "liboracle32.so" if key == 4 else "liboracle64.so"
        l = C.CDLL(os.path.join(_HERE, "_build", name))
        assert l.orc_sizeof_real() == key
        l.orc_gut_bin.restype = C.c_uint64
        l.orc_higher_msb.restype = C.c_uint32
        _libs[key] = l
    return _libs[key]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def round_to_half(a):
    """The reference's fp16 feature I/O as the oracle sees it: values stored through IEEE half (round to nearest even, what
    `.to(torch::kHalf)` of splatRaster.cpp:90-98 / optixTracer.cpp:52-60 and `__float2half` of rayPayload.cuh:176-186 /
    referenceSlangOptix.cu:183-184 do) and read back as fp32 — all arithmetic stays fp32, so a half-mode run of the oracle is an
    fp32 run on rounded coefficients (PARTICLE_FEATURE_HALF) whose image is rounded once at the end (FEATURE_OUTPUT_HALF) and
    handed to the backward in that rounded form (rayPayloadBackward.cuh:50-58, referenceSlangBwdOptix.cu:116-117)."""
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def default_gut_config(**kw) -> GutConfig:
    """configs/render/3dgut.yaml + configs/render/3dgrt.yaml defaults (reference)."""
    cfg = GutConfig(
        particle_kernel_degree=2, particle_kernel_min_response=0.0113, particle_kernel_min_alpha=1.0 / 255.0,
        particle_kernel_max_alpha=0.99, min_transmittance=0.0001, particle_radiance_sph_degree=3,
        enable_hitcounts=1, enable_kernel_timings=0, ut_alpha=1.0, ut_beta=2.0, ut_kappa=0.0,
        ut_in_image_margin_factor=0.1, ut_require_all_sigma_points_valid=0, n_rolling_shutter_iterations=5,
        k_buffer_size=0, global_z_order=1, rect_bounding=1, tight_opacity_bounding=1, tile_based_culling=1)
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def default_grt_config(**kw) -> GrtConfig:
    cfg = GrtConfig(
        particle_kernel_degree=4, particle_kernel_min_response=0.0113, particle_kernel_min_alpha=1.0 / 255.0,
        particle_kernel_max_alpha=0.99, particle_kernel_density_clamping=1, particle_radiance_sph_degree=3,
        enable_normals=0, enable_hitcounts=1, enable_kernel_timings=0, max_hits_per_trace=16)
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


# ------------------------------------------------------------------------------------------
# 3DGUT
# ------------------------------------------------------------------------------------------
def gut_project(cfg, cam, pose_start, pose_end, n_active, density12, sph, dtype=np.float32):
    l = lib(dtype)
    d12, s = _c(density12, dtype), _c(sph, dtype)
    N = d12.shape[0]
    ps, pe = _c(pose_start, dtype), _c(pose_end, dtype)
    out = dict(
        tiles_count=np.zeros(N, np.uint32), proj_pos=np.zeros((N, 2), dtype), conic_opacity=np.zeros((N, 4), dtype),
        extent=np.zeros((N, 2), dtype), depth=np.zeros(N, dtype), rgb=np.zeros((N, 3), dtype),
        visibility=np.zeros(N, np.int32))
    l.orc_gut_project(C.byref(cfg), C.byref(cam), _p(ps), _p(pe), C.c_uint32(N), C.c_int(n_active), _p(d12), _p(s),
                      _p(out["tiles_count"]), _p(out["proj_pos"]), _p(out["conic_opacity"]), _p(out["extent"]),
                      _p(out["depth"]), _p(out["rgb"]), _p(out["visibility"]))
    return out


def gut_bin(cfg, width, height, proj, dtype=np.float32):
    l = lib(dtype)
    N = proj["tiles_count"].shape[0]
    args = (C.byref(cfg), C.c_int(width), C.c_int(height), C.c_uint32(N), _p(proj["tiles_count"]),
            _p(proj["proj_pos"]), _p(proj["conic_opacity"]), _p(proj["extent"]), _p(proj["depth"]))
    total = int(l.orc_gut_bin(*args, None, None, None))
    tiles = ((width + 15) // 16) * ((height + 15) // 16)
    keys = np.zeros(max(total, 1), np.uint64)
    idx = np.zeros(max(total, 1), np.uint32)
    ranges = np.zeros((tiles, 2), np.uint32)
    if total:
        l.orc_gut_bin(*args, _p(keys), _p(idx), _p(ranges))
    return dict(num_intersections=total, sorted_keys=keys[:total], sorted_idx=idx[:total], tile_ranges=ranges)


def gut_forward(cfg, cam, pose_start, pose_end, n_active, density12, sph, ray_o, ray_d, dtype=np.float32, proj=None, lists=None):
    """Full reference forward (SplatRaster::trace semantics incl. output initial values).

    `proj` reuses an earlier projection; `lists` = (sorted_idx u32[I], tile_ranges u32[tiles,2]) composites GIVEN per-tile
    lists instead of the oracle's own binning (the full-size parity tests render the oracle on the lists the GPU built, so
    that the compositing stage is compared on identical hit candidates and the binning stage separately, integer by integer)."""
    l = lib(dtype)
    H, W = cam.height, cam.width
    if proj is None:
        proj = gut_project(cfg, cam, pose_start, pose_end, n_active, density12, sph, dtype)
    if lists is None:
        bins = gut_bin(cfg, W, H, proj, dtype)
    else:
        sidx, rng = np.ascontiguousarray(lists[0], np.uint32), np.ascontiguousarray(lists[1], np.uint32)
        bins = dict(num_intersections=int(sidx.shape[0]), sorted_keys=None, sorted_idx=sidx, tile_ranges=rng)
    fd = np.zeros((H, W, 4), dtype)
    dist = np.full((H, W, 1), 1e6, dtype)
    cnt = np.zeros((H, W, 1), dtype)
    ro, rd = _c(ray_o, dtype).reshape(H, W, 3), _c(ray_d, dtype).reshape(H, W, 3)
    d12 = _c(density12, dtype)
    ps, pe = _c(pose_start, dtype), _c(pose_end, dtype)
    if bins["num_intersections"] > 0:  # gutRenderer.cu:323-325 early return
        r = l.orc_gut_render_fwd(C.byref(cfg), C.c_int(W), C.c_int(H), _p(ps), _p(pe), _p(d12), _p(proj["rgb"]),
                                 _p(bins["sorted_idx"]), _p(bins["tile_ranges"]), _p(ro), _p(rd),
                                 _p(fd), _p(dist), _p(cnt))
        assert r == 0
    return dict(feat_density=fd, hit_distance=dist, hit_count=cnt, visibility=proj["visibility"],
                proj=proj, bins=bins, rays=(ro, rd), density12=d12, sph=_c(sph, dtype), poses=(ps, pe))


NHT_DEFAULT = dict(particle_feature_dim=48, interp_point_dim=12, support=1, activation=2, num_frequencies=1)   # configs/base_gs.yaml:96-103


def nht_ray_feature_dim(nht):
    """threedgrut/model/features.py:152-161."""
    return nht["interp_point_dim"] * (nht["num_frequencies"] * 2 if nht["activation"] == 2 else (nht["num_frequencies"] if nht["activation"] == 1 else 1))


def gut_forward_nht(cfg, cam, pose_start, pose_end, density12, features, ray_o, ray_d, nht=None, dtype=np.float32, lists=None):
    """Reference forward in the neural-harmonic-features configuration (model.feature_type = nht, K = 0): projection and binning as for
    SH (no per-particle radiance), per-hit features interpolated at the canonical intersection -> [H, W, ray_dim + 1]."""
    nht = dict(NHT_DEFAULT, **(nht or {}))
    l = lib(dtype)
    H, W = cam.height, cam.width
    n = len(density12)
    if lists is None:
        proj = gut_project(cfg, cam, pose_start, pose_end, 0, density12, np.zeros((n, 48), np.float32), dtype)
        bins = gut_bin(cfg, W, H, proj, dtype)
    else:
        proj = None
        bins = dict(sorted_idx=np.ascontiguousarray(lists[0], np.uint32), tile_ranges=np.ascontiguousarray(lists[1], np.uint32))
    nr = nht_ray_feature_dim(nht)
    fd = np.zeros((H, W, nr + 1), dtype)
    dist = np.full((H, W, 1), 1e6, dtype)
    cnt = np.zeros((H, W, 1), dtype)
    prm = np.array([nht["particle_feature_dim"], nht["interp_point_dim"], nht["support"], nht["activation"], nht["num_frequencies"]], np.int32)
    ro, rd = _c(ray_o, dtype).reshape(H, W, 3), _c(ray_d, dtype).reshape(H, W, 3)
    rc = l.orc_gut_render_nht_fwd(C.byref(cfg), _p(prm), W, H, _p(_c(pose_start, dtype)), _p(_c(pose_end, dtype)), _p(_c(density12, dtype)),
                                  _p(_c(features, dtype)), _p(bins["sorted_idx"]), _p(bins["tile_ranges"]), _p(ro), _p(rd), _p(fd), _p(dist), _p(cnt))
    assert rc == 0
    return dict(feat_density=fd, hit_distance=dist, hit_count=cnt, sorted_idx=bins["sorted_idx"], tile_ranges=bins["tile_ranges"], proj=proj)


def gut_backward_nht(cfg, cam, pose_start, pose_end, density12, features, ray_o, ray_d, fwd, g_feat_density, g_hit_distance=None, nht=None, dtype=np.float32):
    """Gradients of the nht forward (K = 0): (grad_density12 [N,12], grad_features [N,K])."""
    nht = dict(NHT_DEFAULT, **(nht or {}))
    l = lib(dtype)
    H, W = cam.height, cam.width
    n = len(density12)
    prm = np.array([nht["particle_feature_dim"], nht["interp_point_dim"], nht["support"], nht["activation"], nht["num_frequencies"]], np.int32)
    gd, gf = np.zeros((n, 12), dtype), np.zeros((n, nht["particle_feature_dim"]), dtype)
    gdist = np.zeros((H, W, 1), dtype) if g_hit_distance is None else _c(g_hit_distance, dtype)
    ro, rd = _c(ray_o, dtype).reshape(H, W, 3), _c(ray_d, dtype).reshape(H, W, 3)
    rc = l.orc_gut_render_nht_bwd(C.byref(cfg), _p(prm), W, H, _p(_c(pose_start, dtype)), _p(_c(pose_end, dtype)), _p(_c(density12, dtype)),
                                  _p(_c(features, dtype)), _p(fwd["sorted_idx"]), _p(fwd["tile_ranges"]), _p(ro), _p(rd),
                                  _p(_c(fwd["feat_density"], dtype)), _p(_c(g_feat_density, dtype)), _p(_c(fwd["hit_distance"], dtype)), _p(gdist), _p(gd), _p(gf))
    assert rc == 0
    return gd, gf


def gut_backward(cfg, cam, n_active, fwd, g_feat_density, g_hit_distance, dtype=np.float32):
    """Reference backward (SplatRaster::trace_bwd): returns (grad_density12 [N,12], grad_sph [N,3*ncoef])."""
    l = lib(dtype)
    H, W = cam.height, cam.width
    d12, sph = fwd["density12"], fwd["sph"]
    N = d12.shape[0]
    ps, pe = fwd["poses"]
    ro, rd = fwd["rays"]
    gd = np.zeros((N, 12), dtype)
    grgb = np.zeros((N, 3), dtype)
    gsph = np.zeros_like(sph)
    gfd, gdist = _c(g_feat_density, dtype), _c(g_hit_distance, dtype)
    if fwd["bins"]["num_intersections"] > 0:
        r = l.orc_gut_render_bwd(C.byref(cfg), C.c_int(W), C.c_int(H), _p(ps), _p(pe), _p(d12), _p(fwd["proj"]["rgb"]),
                                 _p(fwd["bins"]["sorted_idx"]), _p(fwd["bins"]["tile_ranges"]), _p(ro), _p(rd),
                                 _p(fwd["feat_density"]), _p(gfd), _p(fwd["hit_distance"]), _p(gdist), _p(gd), _p(grgb))
        assert r == 0
    l.orc_gut_project_bwd(C.byref(cfg), _p(ps), _p(pe), C.c_uint32(N), C.c_int(n_active),
                          _p(fwd["proj"]["tiles_count"]), _p(d12), _p(sph), _p(grgb), _p(gd), _p(gsph))
    return gd, gsph, grgb


# ------------------------------------------------------------------------------------------
# per-hit known-answer entry points (tests/test_oracle_cpu.py against tests/golden/*.npz)
# ------------------------------------------------------------------------------------------
def _real(dtype):
    return C.c_float if np.dtype(dtype).itemsize == 4 else C.c_double


def gut_process_hit_fwd(degree, min_response, min_alpha, max_alpha, ray_o, ray_d, density12, feat3, state5, dtype=np.float32):
    """One 3DGUT hit on a running ray state {T, rgb[3], depth}; returns (accepted, new state)."""
    l, R = lib(dtype), _real(dtype)
    st = _c(state5, dtype).copy()
    T, rgb, dep = st[0:1].copy(), st[1:4].copy(), st[4:5].copy()
    acc = l.orc_gut_process_hit_fwd(C.c_int(degree), R(min_response), R(min_alpha), R(max_alpha), _p(_c(ray_o, dtype)), _p(_c(ray_d, dtype)),
                                    _p(_c(density12, dtype)), _p(_c(feat3, dtype)), _p(T), _p(rgb), _p(dep))
    return int(acc), np.concatenate([T, rgb, dep])


def gut_process_hit_bwd(degree, min_response, min_alpha, max_alpha, min_transmittance, ray_o, ray_d, density12, feat3, state5, fin5,
                        grads5, dtype=np.float32):
    l, R = lib(dtype), _real(dtype)
    st = _c(state5, dtype).copy()
    gd, gf = np.zeros(12, dtype), np.zeros(3, dtype)
    l.orc_gut_process_hit_bwd(C.c_int(degree), R(min_response), R(min_alpha), R(max_alpha), R(min_transmittance), _p(_c(ray_o, dtype)),
                              _p(_c(ray_d, dtype)), _p(_c(density12, dtype)), _p(_c(feat3, dtype)), _p(st), _p(_c(fin5, dtype)),
                              _p(_c(grads5, dtype)), _p(gd), _p(gf))
    return st, gd, gf


def sh_radiance(deg, coeffs48, dir3, clamped=True, dtype=np.float32):
    l = lib(dtype)
    out = np.zeros(3, dtype)
    l.orc_sh_radiance(C.c_int(deg), C.c_int(3), _p(_c(coeffs48, dtype)), _p(_c(dir3, dtype)), C.c_int(int(clamped)), _p(out))
    return out


# ------------------------------------------------------------------------------------------
# 3DGRT
# ------------------------------------------------------------------------------------------
def grt_proxies(cfg, positions, rotations, scales, densities, dtype=np.float32):
    """Instance records {W rows, mu} [N,12], world AABBs [N,6], pruning slack [N], scene AABB [6]."""
    l = lib(dtype)
    pos, rot, scl, dns = _c(positions, dtype), _c(rotations, dtype), _c(scales, dtype), _c(densities, dtype).reshape(-1)
    N = pos.shape[0]
    inst, aabb, slack, scene = np.zeros((N, 12), dtype), np.zeros((N, 6), dtype), np.zeros(N, dtype), np.zeros(6, dtype)
    l.orc_grt_proxies(C.byref(cfg), C.c_uint32(N), _p(pos), _p(rot), _p(scl), _p(dns), _p(inst), _p(aabb), _p(slack), _p(scene))
    return dict(inst=inst, aabb=aabb, slack=slack, scene=scene)


_custom_keepalive = {}


def grt_set_custom_boxes(box8, dtype=np.float32):
    """Registers caller-supplied boxes (e.g. the ones the GPU built: grt_debug_fetch_custom_boxes) for the next traces of this build."""
    b = _c(box8, dtype)
    _custom_keepalive[dtype] = b
    lib(dtype).orc_grt_set_custom_boxes(_p(b))
    return b


def grt_custom_boxes(cfg, density12, dtype=np.float32):
    """render.primitive_type custom: [N,8] {world box min, max, kernelScale^2, 0} (orc_grt_custom_boxes), registered with the library for the
    next traces (orc_grt_set_custom_boxes) and kept alive here."""
    l = lib(dtype)
    d12 = _c(density12, dtype)
    pos, rot, scl, dns = (np.ascontiguousarray(d12[:, 0:3]), np.ascontiguousarray(d12[:, 4:8]), np.ascontiguousarray(d12[:, 8:11]),
                          np.ascontiguousarray(d12[:, 3]))
    box8 = np.zeros((d12.shape[0], 8), dtype)
    l.orc_grt_custom_boxes(C.byref(cfg), C.c_uint32(d12.shape[0]), _p(pos), _p(rot), _p(scl), _p(dns), _p(box8))
    _custom_keepalive[dtype] = box8
    l.orc_grt_set_custom_boxes(_p(box8))
    return box8


_prefilter_keepalive = {}


def grt_set_candidate_prefilter(ranges=None, entries=None, ray_packet=None):
    """Restricts every ray's all-pairs candidate scan to the particles of its packet's list (orc_grt_set_candidate_prefilter; both builds).
    ranges [blocks,2] u32, entries [I] u32, ray_packet [nrays] u32 = packet index of each ray of the NEXT grt_forward / backward calls.
    Call without arguments to clear."""
    for dtype in (np.float32, np.float64):
        l = lib(dtype)
        if ranges is None:
            l.orc_grt_set_candidate_prefilter(None, None, None)
            _prefilter_keepalive.clear()
        else:
            r, e, p = (np.ascontiguousarray(a, np.uint32) for a in (ranges, entries, ray_packet))
            _prefilter_keepalive[dtype] = (r, e, p)
            l.orc_grt_set_candidate_prefilter(_p(r), _p(e), _p(p))


def grt_forward(cfg, density12, sph, sph_deg, min_transmittance, ray_to_world, ray_o, ray_d, inst=None, scene=None, dbg_cap=0,
                dtype=np.float32, box8=None):
    """OptixTracer::trace semantics.  ray_to_world: [3,4]; rays: [H,W,3] in ray space.  `inst` / `scene` may be supplied
    (e.g. the proxies the GPU built) so that hit order can be compared bit-exactly."""
    l, R = lib(dtype), _real(dtype)
    d12, s = _c(density12, dtype), _c(sph, dtype)
    N = d12.shape[0]
    if inst is None:
        pr = grt_proxies(cfg, d12[:, 0:3], d12[:, 4:8], d12[:, 8:11], d12[:, 3], dtype)
        inst, scene = pr["inst"], pr["scene"]
    inst, scene = _c(inst, dtype), _c(scene, dtype)
    if cfg.primitive_type == 5:   # custom primitives: the world boxes (the GPU's, when given - like `inst`)
        box8 = grt_custom_boxes(cfg, d12, dtype) if box8 is None else grt_set_custom_boxes(box8, dtype)
    ro, rd = _c(ray_o, dtype), _c(ray_d, dtype)
    H, W = ro.shape[-3], ro.shape[-2]
    n = H * W
    m = _c(np.asarray(ray_to_world)[:3, :4], dtype)
    out = dict(features=np.zeros((H, W, 3), dtype), density=np.zeros((H, W, 1), dtype), hit_distance=np.zeros((H, W, 2), dtype),
               normals=np.zeros((H, W, 3), dtype), hit_count=np.zeros((H, W, 1), dtype), visibility=np.zeros(N, np.int32))
    dbg_ids = np.full((n, max(dbg_cap, 1)), 0xFFFFFFFF, np.uint32)
    dbg_cnt = np.zeros(n, np.uint32)
    r = l.orc_grt_trace_fwd(C.byref(cfg), C.c_uint32(N), _p(d12), _p(s), C.c_int(sph_deg), R(min_transmittance), _p(inst), _p(scene),
                            _p(m), C.c_uint32(n), _p(ro), _p(rd), _p(out["features"]), _p(out["density"]), _p(out["hit_distance"]),
                            _p(out["normals"]), _p(out["hit_count"]), _p(out["visibility"]),
                            _p(dbg_ids) if dbg_cap else None, _p(dbg_cnt), C.c_uint32(dbg_cap))
    assert r == 0
    out.update(hit_ids=dbg_ids, hit_num=dbg_cnt, inst=inst, scene=scene, density12=d12, sph=s, rays=(ro, rd), ray_to_world=m, box8=box8)
    return out


def _nht_prm(nht):
    nht = dict(NHT_DEFAULT, **(nht or {}))
    return nht, np.array([nht["particle_feature_dim"], nht["interp_point_dim"], nht["support"], nht["activation"], nht["num_frequencies"]], np.int32)


def grt_forward_nht(cfg, density12, features, min_transmittance, ray_to_world, ray_o, ray_d, nht=None, inst=None, scene=None, dbg_cap=0, dtype=np.float32,
                    box8=None):
    """The Slang pipeline with neural harmonic features (referenceSlangOptix.cu): like grt_forward, `features` [N, K] -> out features [H, W, ray_dim]."""
    l, R = lib(dtype), _real(dtype)
    nht, prm = _nht_prm(nht)
    d12, f = _c(density12, dtype), _c(features, dtype)
    N = d12.shape[0]
    if inst is None:
        pr = grt_proxies(cfg, d12[:, 0:3], d12[:, 4:8], d12[:, 8:11], d12[:, 3], dtype)
        inst, scene = pr["inst"], pr["scene"]
    inst, scene = _c(inst, dtype), _c(scene, dtype)
    if cfg.primitive_type == 5:   # custom primitives: the world boxes (the GPU's, when given - like `inst`)
        box8 = grt_custom_boxes(cfg, d12, dtype) if box8 is None else grt_set_custom_boxes(box8, dtype)
    ro, rd = _c(ray_o, dtype), _c(ray_d, dtype)
    H, W = ro.shape[-3], ro.shape[-2]
    n = H * W
    m = _c(np.asarray(ray_to_world)[:3, :4], dtype)
    nr = nht_ray_feature_dim(nht)
    out = dict(features=np.zeros((H, W, nr), dtype), density=np.zeros((H, W, 1), dtype), hit_distance=np.zeros((H, W, 2), dtype),
               hit_count=np.zeros((H, W, 1), dtype), visibility=np.zeros(N, np.int32))
    dbg_ids = np.full((n, max(dbg_cap, 1)), 0xFFFFFFFF, np.uint32)
    dbg_cnt = np.zeros(n, np.uint32)
    r = l.orc_grt_trace_nht_fwd(C.byref(cfg), _p(prm), C.c_uint32(N), _p(d12), _p(f), R(min_transmittance), _p(inst), _p(scene), _p(m), C.c_uint32(n),
                                _p(ro), _p(rd), _p(out["features"]), _p(out["density"]), _p(out["hit_distance"]), _p(out["hit_count"]),
                                _p(out["visibility"]), _p(dbg_ids) if dbg_cap else None, _p(dbg_cnt), C.c_uint32(dbg_cap))
    assert r == 0
    out.update(hit_ids=dbg_ids, hit_num=dbg_cnt, inst=inst, scene=scene, density12=d12, nht_features=f, rays=(ro, rd), ray_to_world=m, nht=nht, box8=box8)
    return out


def grt_backward_nht(cfg, min_transmittance, fwd, g_features, g_density, g_hit_distance=None, dtype=np.float32):
    """referenceSlangBwdOptix.cu with neural harmonic features: (grad_density12 [N,12], grad_features [N,K])."""
    l, R = lib(dtype), _real(dtype)
    nht, prm = _nht_prm(fwd["nht"])
    d12, f = fwd["density12"], fwd["nht_features"]
    N = d12.shape[0]
    ro, rd = fwd["rays"]
    n = ro.shape[-3] * ro.shape[-2]
    gd, gf = np.zeros((N, 12), dtype), np.zeros_like(f)
    gh = np.zeros((n,), dtype) if g_hit_distance is None else _c(g_hit_distance, dtype)
    if cfg.primitive_type == 5:
        grt_set_custom_boxes(fwd["box8"], dtype)
    r = l.orc_grt_trace_nht_bwd(C.byref(cfg), _p(prm), C.c_uint32(N), _p(d12), _p(f), R(min_transmittance), _p(fwd["inst"]), _p(fwd["scene"]),
                                _p(fwd["ray_to_world"]), C.c_uint32(n), _p(ro), _p(rd), _p(_c(fwd["features"], dtype)), _p(_c(fwd["density"], dtype)),
                                _p(_c(fwd["hit_distance"], dtype)), _p(_c(g_features, dtype)), _p(_c(g_density, dtype)), _p(gh), _p(gd), _p(gf))
    assert r == 0
    return gd, gf


def grt_backward(cfg, sph_deg, min_transmittance, fwd, g_features, g_density, g_hit_distance, dtype=np.float32, dbg_cap=0, round_shift=None):
    """OptixTracer::trace_bwd: returns (grad_density12 [N,12], grad_sph [N,3*ncoef]); with dbg_cap > 0 additionally the particles
    each ray's backward program processed, in order: (.., hit_ids [n, dbg_cap], hit_num [n]).  `round_shift`: optional uint8 [n]
    output, 1 where the backward program's hit set differs from the forward's (see orc_grt_trace_bwd)."""
    l, R = lib(dtype), _real(dtype)
    d12, s = fwd["density12"], fwd["sph"]
    N = d12.shape[0]
    ro, rd = fwd["rays"]
    n = ro.shape[-3] * ro.shape[-2]
    gd, gs = np.zeros((N, 12), dtype), np.zeros_like(s)
    if cfg.primitive_type == 5:
        grt_set_custom_boxes(fwd["box8"], dtype)
    dbg_ids = np.full((n, max(dbg_cap, 1)), 0xFFFFFFFF, np.uint32)
    dbg_cnt = np.zeros(n, np.uint32)
    r = l.orc_grt_trace_bwd(C.byref(cfg), C.c_uint32(N), _p(d12), _p(s), C.c_int(sph_deg), R(min_transmittance), _p(fwd["inst"]),
                            _p(fwd["scene"]), _p(fwd["ray_to_world"]), C.c_uint32(n), _p(ro), _p(rd), _p(fwd["features"]),
                            _p(fwd["density"]), _p(fwd["hit_distance"]), _p(_c(g_features, dtype)), _p(_c(g_density, dtype)),
                            _p(_c(g_hit_distance, dtype)), _p(gd), _p(gs), _p(dbg_ids) if dbg_cap else None, _p(dbg_cnt), C.c_uint32(dbg_cap),
                            _p(round_shift) if round_shift is not None else None)
    assert r == 0
    if dbg_cap:
        return gd, gs, dbg_ids, dbg_cnt
    return gd, gs


def grt_process_hit_fwd(degree, min_response, min_alpha, max_alpha, ray_o, ray_d, density12, sph48, sph_deg, with_normal, state8,
                        dtype=np.float32):
    l, R = lib(dtype), _real(dtype)
    st = _c(state8, dtype).copy()
    acc = l.orc_grt_process_hit_fwd(C.c_int(degree), R(min_response), R(min_alpha), R(max_alpha), _p(_c(ray_o, dtype)), _p(_c(ray_d, dtype)),
                                    _p(_c(density12, dtype)), _p(_c(sph48, dtype)), C.c_int(sph_deg), C.c_int(int(with_normal)), _p(st))
    return int(acc), st


def grt_composite_sequence(cfg, min_transmittance, ray_o_world, ray_d_world, density12, sph, sph_deg, particle_ids, dtype=np.float32):
    """The forward ray program's per-hit loop (referenceOptix.cu:141-170) over a GIVEN sequence of particles, in the given order: processHit
    while the transmittance is above the threshold.  Returns (radiance[3], opacity, integrated distance, accepted hits).  For the tests that
    identify order ties: the same hits in another order."""
    l, R = lib(dtype), _real(dtype)
    st = np.zeros(8, dtype)
    st[0] = 1
    ro, rd = _c(ray_o_world, dtype), _c(ray_d_world, dtype)
    d12, s = _c(density12, dtype), _c(sph, dtype)
    n_acc = 0
    for pid in particle_ids:
        if not st[0] > min_transmittance:
            break
        n_acc += int(l.orc_grt_process_hit_fwd_prim(C.c_int(int(cfg.primitive_type)), C.c_int(int(cfg.particle_kernel_degree)), R(cfg.particle_kernel_min_response),
                                                    R(cfg.particle_kernel_min_alpha), R(cfg.particle_kernel_max_alpha), _p(ro), _p(rd),
                                                    _p(np.ascontiguousarray(d12[int(pid)])), _p(np.ascontiguousarray(s[int(pid)])), C.c_int(sph_deg), _p(st)))
    return st[1:4].copy(), float(1 - st[0]), float(st[4]), n_acc


def grt_process_hit_bwd(degree, min_response, min_alpha, max_alpha, min_transmittance, ray_o, ray_d, density12, sph48, sph_deg, state5,
                        fin5, grads5, dtype=np.float32):
    l, R = lib(dtype), _real(dtype)
    st = _c(state5, dtype).copy()
    gd, gs = np.zeros(12, dtype), np.zeros(48, dtype)
    l.orc_grt_process_hit_bwd(C.c_int(degree), R(min_response), R(min_alpha), R(max_alpha), R(min_transmittance), _p(_c(ray_o, dtype)),
                              _p(_c(ray_d, dtype)), _p(_c(density12, dtype)), _p(_c(sph48, dtype)), C.c_int(sph_deg), _p(st),
                              _p(_c(fin5, dtype)), _p(_c(grads5, dtype)), _p(gd), _p(gs))
    return st, gd, gs


def grt_intersect_instance(pray_o, pray_d, tmin, tmax, max_sqdist, dtype=np.float32):
    l, R = lib(dtype), _real(dtype)
    t = np.zeros(1, dtype)
    ok = l.orc_grt_intersect_instance(_p(_c(pray_o, dtype)), _p(_c(pray_d, dtype)), R(tmin), R(tmax), R(max_sqdist), _p(t))
    return int(ok), float(t[0])


def grt_kernel_scale(density, min_response, clamping, degree, dtype=np.float32):
    l, R = lib(dtype), _real(dtype)
    l.orc_grt_kernel_scale.restype = R
    return float(l.orc_grt_kernel_scale(R(density), R(min_response), C.c_int(int(clamping)), R(degree)))


# ---- camera / pose known-answer wrappers (gut_oracle.c: orc_kat_*) ------------------------------------------------
def kat_project_point_with_shutter(cam, pose_start7, pose_end7, n_iter, pos3, tol, dtype=np.float32):
    """projectPointWithShutter (cameraProjections.cuh:218-257) for one world-space point -> (valid, [x, y])."""
    l, R = lib(dtype), _real(dtype)
    out = np.zeros(2, dtype)
    ps, pe, p = _c(pose_start7, dtype), _c(pose_end7, dtype), _c(pos3, dtype)
    ok = l.orc_kat_project_point_with_shutter(C.byref(cam), _p(ps), _p(pe), C.c_int(n_iter), _p(p), R(tol), _p(out))
    return bool(ok), out


def kat_pose_inverse(p7, dtype=np.float32):
    out = np.zeros(7, dtype)
    lib(dtype).orc_kat_pose_inverse(_p(_c(p7, dtype)), _p(out))
    return out


def kat_pose_interpolate(a7, b7, t, dtype=np.float32):
    out = np.zeros(7, dtype)
    lib(dtype).orc_kat_pose_interpolate(_p(_c(a7, dtype)), _p(_c(b7, dtype)), _real(dtype)(t), _p(out))
    return out


def gut_pixel_margins(cfg, cam, fwd, pixel_ids, rel_margin=2e-3, dtype=np.float32):
    """Number of evaluated tile entries of each listed pixel (flat index y*W+x) that lie within `rel_margin` of an accept /
    termination threshold (orc_gut_pixel_margins) — on the lists and rays of the forward `fwd`."""
    l, R = lib(dtype), _real(dtype)
    ids = np.ascontiguousarray(pixel_ids, np.uint32)
    out = np.zeros(ids.shape[0], np.uint32)
    ps, pe = fwd["poses"]
    ro, rd = fwd["rays"]
    if ids.size:
        r = l.orc_gut_pixel_margins(C.byref(cfg), C.c_int(cam.width), C.c_int(cam.height), _p(ps), _p(pe), _p(fwd["density12"]),
                                    _p(fwd["bins"]["sorted_idx"]), _p(fwd["bins"]["tile_ranges"]), _p(ro), _p(rd), C.c_uint32(ids.shape[0]), _p(ids),
                                    R(rel_margin), _p(out))
        assert r == 0
    return out


def gut_pixel_trace(cfg, cam, fwd, pixel, cap=4096, dtype=np.float32):
    """Per-entry trace of one pixel's tile list (orc_gut_pixel_trace): particle ids, compositing alpha, hit distance, signed
    relative margin of the accept test (accepted iff > 0) — for every entry of the tile, without early termination."""
    l = lib(dtype)
    idx = np.zeros(cap, np.uint32)
    alpha, hit_t, margin = np.zeros(cap, dtype), np.zeros(cap, dtype), np.zeros(cap, dtype)
    ps, pe = fwd["poses"]
    ro, rd = fwd["rays"]
    n = l.orc_gut_pixel_trace(C.byref(cfg), C.c_int(cam.width), C.c_int(cam.height), _p(ps), _p(pe), _p(fwd["density12"]),
                              _p(fwd["bins"]["sorted_idx"]), _p(fwd["bins"]["tile_ranges"]), _p(ro), _p(rd), C.c_uint32(int(pixel)),
                              C.c_uint32(cap), _p(idx), _p(alpha), _p(hit_t), _p(margin))
    return dict(idx=idx[:n], alpha=alpha[:n], hit_t=hit_t[:n], margin=margin[:n])


def gut_pixel_trace_nht(cfg, cam, fwd, pixel, cap=4096, dtype=np.float32, nht=None):
    """gut_pixel_trace for the neural-harmonic-features path (orc_gut_pixel_trace_nht): additionally `colour` [n, ray_dim], the feature values
    every entry's hit would blend.  fwd: dict with poses, rays, density12, nht_features, bins (sorted_idx, tile_ranges)."""
    nht = dict(NHT_DEFAULT, **(nht or {}))
    l = lib(dtype)
    nr = nht_ray_feature_dim(nht)
    prm = np.array([nht["particle_feature_dim"], nht["interp_point_dim"], nht["support"], nht["activation"], nht["num_frequencies"]], np.int32)
    idx = np.zeros(cap, np.uint32)
    alpha, hit_t, margin, feat = np.zeros(cap, dtype), np.zeros(cap, dtype), np.zeros(cap, dtype), np.zeros((cap, nr), dtype)
    ps, pe = fwd["poses"]
    ro, rd = fwd["rays"]
    n = l.orc_gut_pixel_trace_nht(C.byref(cfg), _p(prm), C.c_int(cam.width), C.c_int(cam.height), _p(_c(ps, dtype)), _p(_c(pe, dtype)), _p(_c(fwd["density12"], dtype)),
                                  _p(_c(fwd["nht_features"], dtype)), _p(fwd["bins"]["sorted_idx"]), _p(fwd["bins"]["tile_ranges"]), _p(_c(ro, dtype)), _p(_c(rd, dtype)),
                                  C.c_uint32(int(pixel)), C.c_uint32(cap), _p(idx), _p(alpha), _p(hit_t), _p(margin), _p(feat))
    return dict(idx=idx[:n], alpha=alpha[:n], hit_t=hit_t[:n], margin=margin[:n], colour=feat[:n])


class _OrcTexture(C.Structure):
    _fields_ = [("data", C.c_void_p), ("height", C.c_int32), ("width", C.c_int32), ("channels", C.c_int32)]


class _OrcMaterial(C.Structure):
    _fields_ = [("diffuse", _OrcTexture), ("emissive", _OrcTexture), ("metallic_roughness", _OrcTexture), ("normal", _OrcTexture),
                ("diffuse_factor", C.c_float * 4), ("emissive_factor", C.c_float * 3), ("metallic_factor", C.c_float), ("roughness_factor", C.c_float),
                ("transmission_factor", C.c_float), ("ior", C.c_float), ("alpha_cutoff", C.c_float), ("alpha_mode", C.c_uint32)]


class _OrcMesh(C.Structure):
    _fields_ = [("num_vertices", C.c_uint32), ("num_faces", C.c_uint32), ("vertices", C.c_void_p), ("triangles", C.c_void_p),
                ("vertex_normals", C.c_void_p), ("vertex_tangents", C.c_void_p), ("vertex_has_tangents", C.c_void_p), ("prim_type", C.c_void_p),
                ("mat_uv", C.c_void_p), ("mat_id", C.c_void_p), ("refractive_index", C.c_void_p), ("num_materials", C.c_uint32),
                ("materials", C.c_void_p), ("envmap", _OrcTexture), ("envmap_offset", C.c_float * 2)]


def _orc_texture(a, channels, keep):
    if a is None:
        return _OrcTexture(None, 0, 0, channels)
    a = _c(a, np.float32)
    keep.append(a)
    return _OrcTexture(a.ctypes.data, a.shape[0], a.shape[1], channels)


def grt_hybrid(cfg, density12, sph, sph_deg, min_transmittance, ray_to_world, ray_o, ray_d, mesh, opts=0, max_pbr_bounces=8, materials=None,
               envmap=None, envmap_offset=(0.0, 0.0), frame_number=0, inst=None, scene=None, ray_max_t=None, pixel_xy=None, launch_width=0,
               dtype=np.float32):
    """Hybrid mesh + Gaussian path tracing (orc_grt_hybrid_trace; playgroundKernel.cu:39-352, materials.cuh, trace.cuh).
    mesh: dict with vertices [V,3] f32, triangles [F,3] i32, vertex_normals [V,3], prim_type [F] i32 (0 none, 1 mirror, 2 glass, 3 diffuse,
    4 PBR), refractive_index [F] and optionally vertex_tangents [V,3], vertex_has_tangents [V] u8, mat_uv [F,3,2], mat_id [F];
    materials: list of dicts as tests/playground_scenes.material() builds them (factors + optional textures [H,W,C]); envmap [EH,EW,4] or
    None (black).  Rays [H,W,3]; pixel (x, y) seeds the random streams — for a subset of a larger launch pass pixel_xy [H,W,2] (u32 launch
    coordinates) and launch_width.  Returns dict(rgba [H,W,4], last_ray [H,W,6], bounces [H,W])."""
    l, R = lib(dtype), _real(dtype)
    d12, s = _c(density12, dtype), _c(sph, dtype)
    N = d12.shape[0]
    if inst is None:
        pr = grt_proxies(cfg, d12[:, 0:3], d12[:, 4:8], d12[:, 8:11], d12[:, 3], dtype)
        inst, scene = pr["inst"], pr["scene"]
    inst, scene = _c(inst, dtype), _c(scene, dtype)
    ro, rd = _c(ray_o, dtype), _c(ray_d, dtype)
    H, W = ro.shape[-3], ro.shape[-2]
    m = _c(np.asarray(ray_to_world)[:3, :4], dtype)
    V, Fn = np.asarray(mesh["vertices"]).reshape(-1, 3).shape[0], np.asarray(mesh["triangles"]).reshape(-1, 3).shape[0]
    keep = []
    arr = lambda key, dt, default: (keep.append(_c(mesh[key] if mesh.get(key) is not None else default, dt)) or keep[-1])
    v, t = arr("vertices", np.float32, None), arr("triangles", np.int32, None)
    vn = arr("vertex_normals", np.float32, np.zeros((V, 3)))
    vt = arr("vertex_tangents", np.float32, np.zeros((V, 3)))
    vh = arr("vertex_has_tangents", np.uint8, np.zeros(V))
    pt = arr("prim_type", np.int32, None).reshape(-1)
    uv = arr("mat_uv", np.float32, np.zeros((Fn, 3, 2)))
    mid = arr("mat_id", np.int32, np.zeros(Fn)).reshape(-1)
    ri = arr("refractive_index", np.float32, np.ones(Fn)).reshape(-1)
    mats = list(materials) if materials else [dict(diffuse_factor=(0.8, 0.8, 0.8, 1.0), emissive_factor=(0, 0, 0), metallic_factor=0.0, roughness_factor=0.5,
                                                   transmission_factor=0.0, ior=1.5, alpha_mode=0, alpha_cutoff=0.5)]
    marr = (_OrcMaterial * len(mats))()
    for i, mt in enumerate(mats):
        o = marr[i]
        o.diffuse, o.emissive = _orc_texture(mt.get("diffuse_tex"), 4, keep), _orc_texture(mt.get("emissive_tex"), 4, keep)
        o.metallic_roughness, o.normal = _orc_texture(mt.get("metallic_roughness_tex"), 2, keep), _orc_texture(mt.get("normal_tex"), 4, keep)
        for k in range(4):
            o.diffuse_factor[k] = float(mt["diffuse_factor"][k])
        for k in range(3):
            o.emissive_factor[k] = float(mt["emissive_factor"][k])
        o.metallic_factor, o.roughness_factor, o.transmission_factor = float(mt["metallic_factor"]), float(mt["roughness_factor"]), float(mt["transmission_factor"])
        o.ior, o.alpha_cutoff, o.alpha_mode = float(mt["ior"]), float(mt["alpha_cutoff"]), int(mt["alpha_mode"])
    om = _OrcMesh(V, Fn, _p(v), _p(t), _p(vn), _p(vt), _p(vh), _p(pt), _p(uv), _p(mid), _p(ri), len(mats), C.cast(marr, C.c_void_p),
                  _orc_texture(envmap, 4, keep), (C.c_float * 2)(float(envmap_offset[0]), float(envmap_offset[1])))
    rgba, last, bounces = np.zeros((H, W, 4), dtype), np.zeros((H, W, 6), dtype), np.zeros((H, W), np.uint32)
    tmax = _c(ray_max_t, dtype).reshape(-1) if ray_max_t is not None else None
    pxy = np.ascontiguousarray(pixel_xy, np.uint32).reshape(-1, 2) if pixel_xy is not None else None
    r = l.orc_grt_hybrid_trace(C.byref(cfg), C.c_uint32(N), _p(d12), _p(s), C.c_int(sph_deg), R(min_transmittance), _p(inst), _p(scene), _p(m),
                               C.c_uint32(W), C.c_uint32(H), _p(ro), _p(rd), _p(tmax), C.byref(om), C.c_uint32(opts), C.c_uint32(max_pbr_bounces),
                               C.c_uint32(frame_number), _p(pxy), C.c_uint32(launch_width), _p(rgba), _p(last), _p(bounces))
    assert r == 0
    return dict(rgba=rgba, last_ray=last, bounces=bounces)

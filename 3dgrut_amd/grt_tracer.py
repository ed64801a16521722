"""3DGRT renderer plugin: drop-in for `threedgrt_tracer.Tracer` (threedgrt_tracer/tracer.py:50-255).

Same surface — `Tracer(conf)`, `.build_acc(gaussians, rebuild)`, `.render(gaussians, gpu_batch, train, frame_id)`, `.timings`
— and the same autograd contract (differentiable inputs: positions / rotation / scale / density / features; outputs:
features, opacity, hit distance, normals, hit count, visibility).  The OptiX pipeline and the RT cores are replaced by the
LBVH + software traversal of the HIP library behind include/grut_amd.h; PyTorch owns tensors, autograd and the stream.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _abi
from .gut_tracer import (_conf_get, _ptr, _stream_ptr, fused_activations_requested, has_standard_activations, nht_config_from_conf,
                         ray_feature_dim_of)

_DEFAULTS = dict(
    particle_kernel_degree=4, particle_kernel_min_response=0.0113, particle_kernel_min_alpha=1.0 / 255.0,
    particle_kernel_max_alpha=0.99, particle_kernel_density_clamping=True, particle_radiance_sph_degree=3,
    enable_normals=False, enable_hitcounts=True, enable_kernel_timings=False)
_SUPPORTED_PIPELINES = ("reference", "referenceSlang")
_SUPPORTED_PRIMITIVES = ("instances",)


def grt_config_from_conf(conf) -> _abi.GrtConfig:
    """conf.render.* -> GrtConfig (the constructor arguments of OptixTracer, threedgrt_tracer/tracer.py:180-193)."""
    render = _conf_get(conf, "render")
    cfg = _abi.GrtConfig()
    for k, d in _DEFAULTS.items():
        v = _conf_get(render, k, d)
        setattr(cfg, k, type(d)(v) if not isinstance(d, bool) else int(bool(v)))
    cfg.max_hits_per_trace = 16
    # neural harmonic features ride on the Slang pipelines (configs/apps/*_3dgrt_mcmc_nht.yaml: pipeline_type referenceSlang,
    # backward_pipeline_type referenceSlangBwd); SH radiance on the `reference` pipelines
    nht = nht_config_from_conf(conf, cfg)
    pipeline = _conf_get(render, "pipeline_type", "reference")
    # `referenceSlang` with SH radiance: the Slang programs (referenceSlangOptix.cu + gaussianParticles.slang / shRadiativeParticles.slang)
    # integrate the same function as the hand-written ones — same hit test, response, alpha clamps, weights and radiance — in another
    # rounding order; both names are served by the same kernels (the SH path is pinned by the `reference` programs only)
    allowed = _SUPPORTED_PIPELINES
    # `barycentricSurfels` (barycentricSurfelsOptix.cu, round 6): the surfel forward pipeline - trisurfel proxies, ten hits per trace, the
    # response from the hit triangle's barycentrics.  FORWARD ONLY: the reference ships no backward program for it (the constructor's default
    # backward name `barycentricSurfelsBwd` has no file, optixTracer.cpp:311-314); backward() raises here
    if pipeline == "barycentricSurfels":
        if _conf_get(render, "primitive_type", "instances") != "trisurfel":
            raise NotImplementedError("3dgrut_amd: render.pipeline_type='barycentricSurfels' needs render.primitive_type='trisurfel' "
                                      "(the program reads the trisurfel kernel's per-particle rows, optixTracer.cpp:735-748)")
        if nht:
            raise NotImplementedError("3dgrut_amd: render.pipeline_type='barycentricSurfels' integrates SH radiance only")
        cfg.pipeline_type = 1
        cfg.max_hits_per_trace = 10
    elif pipeline not in allowed:
        raise NotImplementedError(f"3dgrut_amd: render.pipeline_type={pipeline!r} is not supported (only {allowed + ('barycentricSurfels',)})")
    bwd_pipeline = _conf_get(render, "backward_pipeline_type", pipeline + "Bwd")
    if bwd_pipeline not in tuple(p + "Bwd" for p in allowed) + (("barycentricSurfelsBwd",) if pipeline == "barycentricSurfels" else ()):
        raise NotImplementedError(f"3dgrut_amd: render.backward_pipeline_type={bwd_pipeline!r} is not supported "
                                  f"(only {tuple(p + 'Bwd' for p in allowed)})")
    prim = _conf_get(render, "primitive_type", "instances")
    if prim not in _abi.GRT_PRIMITIVES:
        raise NotImplementedError(f"3dgrut_amd: render.primitive_type={prim!r} is not supported (provided: {tuple(_abi.GRT_PRIMITIVES)})")
    if nht and prim == "trisurfel":
        # (the feature path walks the trace kernel's hit log and evaluates the features at each hit's canonical intersection: it does not care
        # which candidate test ordered the log - instances, the closed mesh proxies, trihexa and sphere (round 6: proxy -> particle at the
        # per-hit sites), custom (round 6: with the Slang pipeline's own test, which reports an unsigned distance, gaussianParticles.slang:489-523).
        # The surfel variant blends at the ray's crossing of the surfel's plane: not built)
        raise NotImplementedError("3dgrut_amd: neural harmonic features are not provided with primitive_type trisurfel (every other proxy is): the Slang "
                                  "surfel mode (scale.z = 1e-6) leaves the feature lookup point's z to the generated code's rounding order - DESIGN.md 7c")
    cfg.primitive_type = _abi.GRT_PRIMITIVES[prim]
    # fp16 feature I/O (setup_3dgrt.py:41-44): run-time switches here, compile-time macros in the reference
    cfg.particle_feature_half = int(bool(_conf_get(render, "particle_feature_half", False)))
    cfg.feature_output_half = int(bool(_conf_get(render, "feature_output_half", False)))
    return cfg


def features_for_kernel(cfg, sph):
    """optixTracer.cpp:52-60 particleFeaturesKernelTensor: the SH buffer the kernels read (half with PARTICLE_FEATURE_HALF)."""
    sph = sph.contiguous()
    if cfg.particle_feature_half and sph.dtype != torch.float16:
        sph = sph.to(torch.float16)
    return sph


class _GrtNative:
    """Owns the C handle (role of lib3dgrt_cc.OptixTracer)."""

    def __init__(self, cfg: _abi.GrtConfig):
        if not torch.cuda.is_available():
            raise RuntimeError("3dgrut_amd.Tracer needs a ROCm GPU (there is no CPU fallback)")
        torch.zeros(1, device="cuda")
        self.lib = _abi.load_library()
        self.cfg = cfg
        self.handle = C.c_void_p()
        _abi.check(self.lib.grt_create(C.byref(cfg), C.byref(self.handle)), "grt_create")
        self.ncoef = (cfg.particle_radiance_sph_degree + 1) ** 2
        self._timings = {}

    def __del__(self):
        try:
            if getattr(self, "handle", None) and self.handle.value:
                self.lib.grt_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:
            pass

    def build_bvh(self, pos, rot, scl, dns, rebuild, allow_update):
        _abi.check(self.lib.grt_build_bvh(self.handle, _stream_ptr(pos.device), pos.shape[0], _ptr(pos), _ptr(rot), _ptr(scl), _ptr(dns),
                                          int(bool(rebuild)), int(bool(allow_update))), "grt_build_bvh")

    def make_frame(self, frame_id, sph_degree, min_transmittance, n, height, width, ray_to_world) -> _abi.GrtFrame:
        f = _abi.GrtFrame()
        f.frame_id, f.sph_degree, f.min_transmittance = int(frame_id) & 0xFFFFFFFF, int(sph_degree), float(min_transmittance)
        f.num_particles, f.width, f.height = int(n), int(width), int(height)
        m = ray_to_world.detach().reshape(-1, 4, 4)[0]
        if m.is_cuda:   # the pose stays on the device (the reference copies it to the host per call: rayToWorld.cpu(), optixTracer.cpp:931)
            md = m.to(torch.float32).clone()   # a snapshot, like the reference's .cpu(): the backward reads it again
            f.device_ray_to_world = md.data_ptr()
            f._keepalive = md   # the frame is kept for the backward: so is the matrix it points to
        else:
            m = m.to(torch.float32)
            for r in range(3):
                for c in range(4):
                    f.ray_to_world[4 * r + c] = float(m[r, c])
        return f

    def trace(self, frame, particle_density, particle_sph, ray_ori, ray_dir, hit_capacity=0):
        dev = ray_ori.device
        H, W, N = frame.height, frame.width, frame.num_particles
        opts = dict(dtype=torch.float32, device=dev)
        # (optixTracer.cpp:903-909: the integrated features are a half tensor with FEATURE_OUTPUT_HALF)
        feat = torch.zeros((1, H, W, ray_feature_dim_of(self.cfg)), dtype=torch.float16 if self.cfg.feature_output_half else torch.float32, device=dev)
        dns = torch.zeros((1, H, W, 1), **opts)
        hit = torch.zeros((1, H, W, 2), **opts)
        nrm = torch.zeros((1, H, W, 3), **opts)
        cnt = torch.zeros((1, H, W, 1), **opts)
        vis = torch.zeros((N, 1), dtype=torch.int32, device=dev)
        args = (self.handle, _stream_ptr(dev), C.byref(frame), _ptr(particle_density), _ptr(particle_sph), _ptr(ray_ori), _ptr(ray_dir),
                _ptr(feat), _ptr(dns), _ptr(hit), _ptr(nrm), _ptr(cnt), _ptr(vis))
        if hit_capacity:
            ids = torch.full((H * W, hit_capacity), -1, dtype=torch.int32, device=dev)
            num = torch.zeros((H * W,), dtype=torch.int32, device=dev)
            _abi.check(self.lib.grt_debug_forward_hits(*args, _ptr(ids), _ptr(num), hit_capacity), "grt_debug_forward_hits")
            return feat, dns, hit, nrm, cnt, vis.view(torch.float32), ids, num
        _abi.check(self.lib.grt_forward(*args), "grt_forward")
        return feat, dns, hit, nrm, cnt, vis.view(torch.float32)

    def trace_bwd(self, frame, particle_density, particle_sph, ray_ori, ray_dir, feat, dns, hit, nrm, g_feat, g_dns, g_hit, g_nrm):
        dev = ray_ori.device
        g_density = torch.zeros_like(particle_density)   # accumulated with atomics (optixTracer.cpp:982-983)
        g_sph = torch.zeros_like(particle_sph, dtype=torch.float32)   # (fp32 also for half coefficients)
        _abi.check(self.lib.grt_backward(self.handle, _stream_ptr(dev), C.byref(frame), _ptr(particle_density), _ptr(particle_sph),
                                         _ptr(ray_ori), _ptr(ray_dir), _ptr(feat), _ptr(dns), _ptr(hit), _ptr(nrm), _ptr(g_feat), _ptr(g_dns),
                                         _ptr(g_hit), _ptr(g_nrm), _ptr(g_density), _ptr(g_sph)), "grt_backward")
        return g_density, g_sph

    def collect_times(self):
        if not self.cfg.enable_kernel_timings:
            return {}
        f, b, bu = C.c_float(-1), C.c_float(-1), C.c_float(-1)
        _abi.check(self.lib.grt_timings(self.handle, C.byref(f), C.byref(b), C.byref(bu)), "grt_timings")
        for k, v in (("forward_render", f.value), ("backward_render", b.value), ("build_bvh", bu.value)):
            if v >= 0:
                self._timings[k] = v
        return dict(self._timings)

    def trim(self):
        """grt_trim: hand all scratch back to the allocator (torch's pool by default); the next frame allocates afresh and the BVH must be rebuilt."""
        _abi.check(self.lib.grt_trim(self.handle), "grt_trim")

    def stats(self) -> _abi.GrtStats:
        s = _abi.GrtStats()
        _abi.check(self.lib.grt_stats(self.handle, C.byref(s)), "grt_stats")
        return s

    def backward_signature(self, num_rays, device):
        """Parity aid: from now on every backward also records, per ray, how many hits it differentiated and an order-independent
        signature of the particles (two tensors, filled by the next backward; call with num_rays = 0 to stop)."""
        if not num_rays:
            self._sig = None
            _abi.check(self.lib.grt_debug_backward_signature(self.handle, None, None), "grt_debug_backward_signature")
            return None
        sig = torch.zeros(num_rays, dtype=torch.int64, device=device)
        cnt = torch.zeros(num_rays, dtype=torch.int32, device=device)
        self._sig = (sig, cnt)   # kept alive while the library writes into them
        _abi.check(self.lib.grt_debug_backward_signature(self.handle, _ptr(sig), _ptr(cnt)), "grt_debug_backward_signature")
        return sig, cnt

    def fetch_lists(self, width, height, device):
        """(ranges [blocks,2], entries [I]) int32 tensors holding the packet lists of the last train-mode forward (grt_debug_fetch_lists)."""
        n = int(self.stats().list_entries)
        blocks = ((width + 7) // 8) * ((height + 7) // 8)
        ranges = torch.zeros((blocks, 2), dtype=torch.int32, device=device)
        entries = torch.zeros(max(n, 1), dtype=torch.int32, device=device)
        _abi.check(self.lib.grt_debug_fetch_lists(self.handle, _stream_ptr(device), _ptr(ranges), _ptr(entries), n), "grt_debug_fetch_lists")
        return ranges, entries[:n]

    def instances(self, n, device):
        out = torch.zeros((n, 12), dtype=torch.float32, device=device)
        _abi.check(self.lib.grt_debug_fetch_instances(self.handle, _stream_ptr(device), _ptr(out)), "grt_debug_fetch_instances")
        return out


    def custom_boxes(self, n, device):
        """primitive_type custom: [n,8] world boxes + kernelScale^2 of the last build (grt_debug_fetch_custom_boxes)."""
        out = torch.zeros((n, 8), dtype=torch.float32, device=device)
        _abi.check(self.lib.grt_debug_fetch_custom_boxes(self.handle, _stream_ptr(device), _ptr(out)), "grt_debug_fetch_custom_boxes")
        return out


class Tracer:
    class _Autograd(torch.autograd.Function):
        @staticmethod
        def forward(ctx, native, frame, ray_ori, ray_dir, mog_pos, mog_rot, mog_scl, mog_dns, mog_sph, raw=False):
            if raw:   # RAW rotation / scale / density, activated inside the packing kernel
                particle_density = _abi.activate_pack(mog_pos, mog_dns, mog_rot, mog_scl)
            else:
                particle_density = _abi.pack_particles(mog_pos, mog_dns, mog_rot, mog_scl)  # [N,12] rows, one pass (tracer.py's torch.cat)
            ctx.raw = (mog_dns, mog_rot, mog_scl) if raw else None
            particle_sph = features_for_kernel(native.cfg, mog_sph)
            feat, dns, hit, nrm, cnt, vis = native.trace(frame, particle_density, particle_sph, ray_ori, ray_dir)
            ctx.save_for_backward(ray_ori, ray_dir, feat, dns, hit, nrm, particle_density, particle_sph)
            if feat.dtype != torch.float32:   # fp32 to the caller, the half image stays in the context for the backward (tracer.py:98)
                feat = feat.float()
            ctx.native, ctx.frame = native, frame
            ctx.mark_non_differentiable(cnt, vis)
            ctx.set_materialize_grads(False)
            # only the integrated hit distance leaves the op (threedgrt_tracer/tracer.py:100)
            return feat, dns, hit[:, :, :, 0:1], nrm, cnt, vis

        @staticmethod
        def backward(ctx, g_feat, g_dns, g_hit, g_nrm, _g_cnt, _g_vis):
            ray_ori, ray_dir, feat, dns, hit, nrm, particle_density, particle_sph = ctx.saved_tensors
            g_feat = torch.zeros_like(feat, dtype=torch.float32) if g_feat is None else g_feat.contiguous()
            g_dns = torch.zeros_like(dns) if g_dns is None else g_dns.contiguous()
            g_hit = None if g_hit is None else g_hit.contiguous()
            g_density, g_sph = ctx.native.trace_bwd(ctx.frame, particle_density, particle_sph, ray_ori, ray_dir, feat, dns, hit, nrm,
                                                    g_feat, g_dns, g_hit, None if g_nrm is None else g_nrm.contiguous())
            if ctx.raw is not None:
                g_pos, g_d, g_rot, g_scl = _abi.activate_pack_backward(*ctx.raw, g_density)
                return None, None, None, None, g_pos, g_rot, g_scl, g_d, g_sph, None
            g_pos, g_d, g_rot, g_scl = _abi.unpack_particle_grads(g_density)
            return None, None, None, None, g_pos, g_rot, g_scl, g_d, g_sph, None

    def __init__(self, conf):
        self.device = "cuda"
        self.conf = conf
        self.num_update_bvh = 0
        render = _conf_get(conf, "render")
        self._clamping = bool(_conf_get(render, "particle_kernel_density_clamping", True))
        self._max_updates = int(_conf_get(render, "max_consecutive_bvh_update", 15))
        self._min_transmittance = float(_conf_get(render, "min_transmittance", 0.001))
        # render.backward_hit_replay (a key of this plugin, default true): the backward replays the hits the forward logged instead
        # of traversing the BVH again (4x faster).  Replaying is the reference's backward program except on ~0.3 % of the rays:
        # where the end-of-ray clip of that program (referenceBwdOptix.cu:123-128) removes a hit, every later k = 16 round
        # boundary moves.  The forward flags those rays and the backward re-derives their rounds exactly, so both settings give
        # the reference's backward; false = re-derive every ray (tests/parity_util.py: grt_full_parity checks both).
        self._replay = bool(_conf_get(render, "backward_hit_replay", True))
        self.tracer_wrapper = _GrtNative(grt_config_from_conf(conf))
        self._fused_activations = fused_activations_requested(conf)

    @property
    def timings(self):
        return self.tracer_wrapper.collect_times()

    def build_acc(self, gaussians, rebuild=True):
        """threedgrt_tracer/tracer.py:198-216: refits are allowed only without density clamping."""
        allow_update = (self._max_updates > 1) and not self._clamping
        rebuild_bvh = bool(rebuild) or self._clamping or self.num_update_bvh >= self._max_updates
        with torch.no_grad():
            if hasattr(gaussians, "rotation_activation"):
                rot = gaussians.rotation_activation(gaussians.rotation)
                scl = gaussians.scale_activation(gaussians.scale)
                dns = gaussians.density_activation(gaussians.density)
            else:
                rot, scl, dns = gaussians.get_rotation(), gaussians.get_scale(), gaussians.get_density()
            self.tracer_wrapper.build_bvh(gaussians.positions.detach().view(-1, 3).contiguous().float(), rot.detach().view(-1, 4).contiguous().float(),
                                          scl.detach().view(-1, 3).contiguous().float(), dns.detach().view(-1, 1).contiguous().float(),
                                          rebuild_bvh, allow_update)
        self.num_update_bvh = 0 if rebuild_bvh else self.num_update_bvh + 1

    def render(self, gaussians, gpu_batch, train=False, frame_id=0):
        get = (lambda k: gpu_batch[k]) if isinstance(gpu_batch, dict) else (lambda k: getattr(gpu_batch, k))
        rays_o, rays_d, T = get("rays_ori"), get("rays_dir"), get("T_to_world")
        H, W = int(rays_o.shape[1]), int(rays_o.shape[2])
        native = self.tracer_wrapper
        feats = gaussians.get_features()
        want = native.cfg.particle_feature_dim if native.cfg.feature_transform_type else 3 * native.ncoef
        if feats.shape[1] != want:
            raise ValueError(f"features have {feats.shape[1]} columns, expected {want}")
        frame = native.make_frame(frame_id, gaussians.n_active_features, self._min_transmittance, gaussians.num_gaussians, H, W, T)
        frame.keep_hits_for_backward = int(bool(train) and torch.is_grad_enabled() and self._replay)
        if self._fused_activations and has_standard_activations(gaussians):
            pred_features, pred_opacity, pred_dist, pred_normals, hits_count, mog_visibility = Tracer._Autograd.apply(
                native, frame, rays_o.contiguous().float(), rays_d.contiguous().float(), gaussians.positions.contiguous(),
                gaussians.rotation.contiguous(), gaussians.scale.contiguous(), gaussians.density.contiguous(), feats.contiguous(), True)
        else:
            pred_features, pred_opacity, pred_dist, pred_normals, hits_count, mog_visibility = Tracer._Autograd.apply(
                native, frame, rays_o.contiguous().float(), rays_d.contiguous().float(), gaussians.positions.contiguous(),
                gaussians.get_rotation().contiguous(), gaussians.get_scale().contiguous(), gaussians.get_density().contiguous(), feats.contiguous())
        timings = native.collect_times()
        return {
            "pred_features": pred_features,
            "pred_opacity": pred_opacity,
            "pred_dist": pred_dist,
            "pred_normals": torch.nn.functional.normalize(pred_normals, dim=3),
            "hits_count": hits_count,
            "frame_time_ms": timings.get("forward_render", 0.0),
            "mog_visibility": mog_visibility,
        }

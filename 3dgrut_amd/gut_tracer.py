"""3DGUT renderer plugin: drop-in for `threedgut_tracer.Tracer` (threedgut_tracer/tracer.py:158-349).

Same surface — `Tracer(conf)`, `.build_acc(gaussians, rebuild)`, `.render(gaussians, gpu_batch, train, frame_id)`,
`.timings` — same autograd contract (differentiable inputs: positions / rotation / scale / density / features;
outputs: feature+opacity image, hit distance, hit count, visibility), but the device work is the HIP library
behind include/grut_amd.h.  PyTorch only owns tensors, the autograd graph and the current stream.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _abi
from .camera import camera_from_batch

_FALSE_DEFAULTS = dict(
    particle_kernel_degree=2, particle_kernel_min_response=0.0113, particle_kernel_min_alpha=1.0 / 255.0,
    particle_kernel_max_alpha=0.99, min_transmittance=0.0001, particle_radiance_sph_degree=3,
    enable_hitcounts=True, enable_kernel_timings=False)
_SPLAT_DEFAULTS = dict(
    ut_alpha=1.0, ut_beta=2.0, ut_kappa=0.0, ut_in_image_margin_factor=0.1, ut_require_all_sigma_points_valid=False,
    n_rolling_shutter_iterations=5, k_buffer_size=0, global_z_order=True, rect_bounding=True,
    tight_opacity_bounding=True, tile_based_culling=True)


def _conf_get(node, name, default=None):
    """OmegaConf DictConfig, SimpleNamespace or plain dict — attribute or item access."""
    if node is None:
        return default
    if isinstance(node, dict):
        return node.get(name, default)
    try:
        v = getattr(node, name)
        return default if v is None else v
    except (AttributeError, KeyError):
        try:
            return node[name]
        except Exception:
            return default


def fused_activations_requested(conf) -> bool:
    """`render.fused_activations: true` (a key of this plugin, absent from the reference's configs) or
    GRUT_FUSED_ACTIVATIONS=1: take the model's RAW density / rotation / scale and apply sigmoid / normalize / exp inside
    the packing kernel (and their chain rule inside the backward), SURVEY.md §8f-3."""
    return bool(_conf_get(_conf_get(conf, "render"), "fused_activations", False)) or bool(os.environ.get("GRUT_FUSED_ACTIVATIONS"))


def has_standard_activations(gaussians) -> bool:
    """The model stores raw parameters and activates them with exactly the functions the fused kernels implement
    (threedgrut/model/model.py:237-241 with configs/base_gs.yaml:77-78; utils/misc.py:44-49)."""
    return (all(hasattr(gaussians, a) for a in ("density", "rotation", "scale", "density_activation", "scale_activation", "rotation_activation"))
            and gaussians.density_activation is torch.sigmoid and gaussians.scale_activation is torch.exp
            and gaussians.rotation_activation is torch.nn.functional.normalize)


def nht_config_from_conf(conf, cfg) -> bool:
    """model.feature_type = nht -> the six feature fields of GutConfig / GrtConfig (the macro set of threedgrut/model/features.py:133-175,
    setup_3dgut.py:47-57 / setup_3dgrt.py); returns whether neural harmonic features are on."""
    model = _conf_get(conf, "model")
    if str(_conf_get(model, "feature_type", "sh")).lower() != "nht":
        return False
    nf = _conf_get(model, "nht_features")
    act = _conf_get(nf, "activation")
    interp = str(_conf_get(nf, "interpolation_type", "none")).lower()
    if interp not in ("none", "barycentric"):
        raise NotImplementedError(f"3dgrut_amd: nht_features.interpolation_type={interp!r} (the reference supports none / barycentric)")
    points = 4 if interp == "barycentric" else 1
    dim = int(_conf_get(nf, "dim", 48))
    if dim % points:
        raise ValueError(f"nht_features.dim = {dim} is not divisible by the {points} interpolation points of interpolation_type={interp!r}")
    cfg.feature_transform_type = 1
    cfg.particle_feature_dim = dim
    cfg.interp_point_feature_dim = dim // points
    cfg.feature_interpolation_support = 1 if points == 4 else 0
    # defaults as threedgrut/model/features.py:62-77, 97-110: no activation node / no type -> none; none and relu have one "frequency"
    act_name = str(_conf_get(act, "type", "none") if act is not None else "none").lower()
    if act_name not in ("none", "siren", "sincos", "relu"):
        raise ValueError(f"Unknown nht_features.activation.type: {act_name}")
    cfg.feature_activation_type = {"none": 0, "siren": 1, "sincos": 2, "relu": 3}[act_name]
    cfg.feature_activation_num_frequencies = 1 if act_name in ("none", "relu") else int(_conf_get(act, "num_frequencies", 1))
    return True


def ray_feature_dim_of(cfg) -> int:
    if not cfg.feature_transform_type:
        return 3
    a, n = cfg.feature_activation_type, cfg.feature_activation_num_frequencies
    return cfg.interp_point_feature_dim * (2 * n if a == 2 else (n if a == 1 else 1))


def gut_config_from_conf(conf) -> _abi.GutConfig:
    """conf.render.* -> GutConfig (the keys setup_3dgut.py:41-95 turns into -D macros)."""
    render = _conf_get(conf, "render")
    splat = _conf_get(render, "splat")
    cfg = _abi.GutConfig()
    for k, d in _FALSE_DEFAULTS.items():
        v = _conf_get(render, k, d)
        setattr(cfg, k, type(d)(v) if not isinstance(d, bool) else int(bool(v)))
    for k, d in _SPLAT_DEFAULTS.items():
        v = _conf_get(splat, k, d)
        setattr(cfg, k, type(d)(v) if not isinstance(d, bool) else int(bool(v)))
    nht_config_from_conf(conf, cfg)
    # fp16 feature I/O (setup_3dgut.py:60-61): run-time switches here, compile-time macros in the reference
    cfg.particle_feature_half = int(bool(_conf_get(render, "particle_feature_half", False)))
    cfg.feature_output_half = int(bool(_conf_get(render, "feature_output_half", False)))
    if _conf_get(splat, "fine_grained_load_balancing", False):
        # Accepted and ignored: the flag selects the reference's warp-per-pixel forward (renderBalanced), whose images differ
        # from its sequential kernel's only in the opacity of rays that end on the transmittance threshold, by < 1e-4
        # (measured on the reference's own two kernels, tests/test_oracle_cpu.py::test_reference_balanced_forward_...).
        # This renderer always computes the sequential result; its work split is already strip-granular.
        pass
    return cfg


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _uninitialised(shape, **opts):
    """torch.empty, or NaN / -1 poison with GRUT_POISON_OUTPUTS=1 (tests: proves that the library overwrites everything)."""
    if os.environ.get("GRUT_POISON_OUTPUTS"):
        return torch.full(shape, float("nan") if opts.get("dtype") == torch.float32 else -1, **opts)
    return torch.empty(shape, **opts)


class _GutNative:
    """Owns the C handle (role of lib3dgut_cc.SplatRaster)."""

    def __init__(self, cfg: _abi.GutConfig):
        if not torch.cuda.is_available():
            raise RuntimeError("3dgrut_amd.Tracer needs a ROCm GPU (there is no CPU fallback)")
        torch.zeros(1, device="cuda")  # force context creation (tracer.py:292)
        self.lib = _abi.load_library()
        self.cfg = cfg
        self.handle = C.c_void_p()
        _abi.check(self.lib.gut_create(C.byref(cfg), C.byref(self.handle)), "gut_create")
        self.ncoef = (cfg.particle_radiance_sph_degree + 1) ** 2
        self._timings = {}  # last collected averages persist like SplatRaster::m_timings (splatRaster.cpp:352-382)

    def __del__(self):
        try:
            if getattr(self, "handle", None) and self.handle.value:
                self.lib.gut_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:
            pass

    def make_frame(self, frame_id, n_active, n, height, width, cam, pose_start, pose_end) -> _abi.GutFrame:
        f = _abi.GutFrame()
        f.frame_id, f.n_active_features, f.num_particles = int(frame_id) & 0xFFFFFFFF, int(n_active), int(n)
        f.width, f.height = int(width), int(height)
        f.camera = cam
        for i in range(7):
            f.pose_start[i] = float(pose_start[i])
            f.pose_end[i] = float(pose_end[i])
        return f

    def trace(self, frame, particle_density, particle_sph, ray_ori, ray_dir):
        dev = ray_ori.device
        H, W = frame.height, frame.width
        N = frame.num_particles
        opts = dict(dtype=torch.float32, device=dev)
        half_out = bool(self.cfg.feature_output_half)
        fd_opts = dict(dtype=torch.float16 if half_out else torch.float32, device=dev)   # (splatRaster.cpp:198-206)
        if N == 0:  # nothing is launched: the outputs are the reference's initial values (splatRaster.cpp:211-215)
            out_fd = torch.zeros((H, W, 4), **fd_opts)
            out_dist = torch.full((H, W, 1), 1e6, **opts)
            out_cnt = torch.zeros((H, W, 1), **opts)
            vis_i32 = torch.zeros((N, 1), dtype=torch.int32, device=dev)
        else:
            # every pixel / particle is written by the library (tiles cover the image; dead rays store their initial
            # values), so no fill passes: the reference pays four torch::zeros / full per frame here
            out_fd = _uninitialised((H, W, 4), **fd_opts)
            out_dist = _uninitialised((H, W, 1), **opts)
            out_cnt = _uninitialised((H, W, 1), **opts) if self.cfg.enable_hitcounts else torch.zeros((H, W, 1), **opts)
            vis_i32 = _uninitialised((N, 1), dtype=torch.int32, device=dev)
        # `pred_features` / `pred_opacity` as contiguous tensors of their own, like the reference's `.contiguous()` slices
        # (tracer.py:334-337), written by the compositing kernel next to the packed image the backward reads
        if half_out:   # the caller's fp32 tensors are converted from the half image afterwards (tracer.py:214-215 `.float()`)
            out_feat = out_opa = None
        elif N == 0:
            out_feat, out_opa = torch.zeros((H, W, 3), **opts), torch.zeros((H, W, 1), **opts)
        else:
            out_feat, out_opa = _uninitialised((H, W, 3), **opts), _uninitialised((H, W, 1), **opts)
        frame.out_features, frame.out_opacity = (None, None) if half_out else (out_feat.data_ptr(), out_opa.data_ptr())
        _abi.check(self.lib.gut_forward(self.handle, _stream_ptr(dev), C.byref(frame), _ptr(particle_density), _ptr(particle_sph),
                                        _ptr(ray_ori), _ptr(ray_dir), _ptr(out_fd), _ptr(out_dist), _ptr(out_cnt), _ptr(vis_i32)),
                   "gut_forward")
        # the reference returns a float tensor holding the int bit pattern (splatRaster.cpp:215,249); consumers call .bool()
        frame.out_features, frame.out_opacity = None, None   # the frame outlives this call in the autograd context
        if half_out:
            f32 = out_fd.float()
            out_feat, out_opa = f32[..., :3].contiguous(), f32[..., 3:].contiguous()
        return out_fd, out_dist, out_cnt, vis_i32.view(torch.float32), out_feat, out_opa

    @property
    def ray_feature_dim(self):
        return ray_feature_dim_of(self.cfg)

    def trace_nht(self, frame, particle_density, particle_features, ray_ori, ray_dir):
        """Forward of the neural-harmonic-features configuration (GutConfig::feature_transform_type 1): [H,W,ray_dim+1] features + opacity."""
        dev = ray_ori.device
        H, W, N = frame.height, frame.width, frame.num_particles
        nr = self.ray_feature_dim
        fd = torch.zeros((H, W, nr + 1), dtype=torch.float16 if self.cfg.feature_output_half else torch.float32, device=dev)
        dist = torch.full((H, W, 1), 1e6, dtype=torch.float32, device=dev)
        cnt = torch.zeros((H, W, 1), dtype=torch.float32, device=dev)
        vis = torch.zeros((N, 1), dtype=torch.int32, device=dev)
        if self.cfg.particle_feature_half and particle_features.dtype != torch.float16:
            particle_features = particle_features.to(torch.float16)
        _abi.check(self.lib.gut_forward(self.handle, _stream_ptr(dev), C.byref(frame), _ptr(particle_density), _ptr(particle_features.contiguous()),
                                        _ptr(ray_ori), _ptr(ray_dir), _ptr(fd), _ptr(dist), _ptr(cnt), _ptr(vis)), "gut_forward")
        return fd, dist, cnt, vis.view(torch.float32), particle_features

    def trace_bwd_nht(self, frame, particle_density, particle_features, ray_ori, ray_dir, fd, g_fd, dist, g_dist):
        """gut_backward of the nht configuration: (packed gradient [N,12], feature-buffer gradient [N,K] fp32), both fully written."""
        dev = ray_ori.device
        g_density = torch.empty_like(particle_density)
        g_feat = torch.empty(particle_features.shape, dtype=torch.float32, device=dev)
        _abi.check(self.lib.gut_backward(self.handle, _stream_ptr(dev), C.byref(frame), _ptr(particle_density), _ptr(particle_features),
                                         _ptr(ray_ori), _ptr(ray_dir), _ptr(fd), _ptr(g_fd), _ptr(dist), _ptr(g_dist),
                                         _ptr(g_density), _ptr(g_feat)), "gut_backward")
        return g_density, g_feat

    def trace_bwd(self, frame, particle_density, particle_sph, ray_ori, ray_dir, fd, g_fd, dist, g_dist):
        dev = ray_ori.device
        g_density = torch.empty_like(particle_density)  # both fully overwritten by the gradient-finalisation kernel
        g_sph = torch.empty_like(particle_sph, dtype=torch.float32)   # (fp32 also for half coefficients, splatRaster.cpp:300-312)
        _abi.check(self.lib.gut_backward(self.handle, _stream_ptr(dev), C.byref(frame), _ptr(particle_density), _ptr(particle_sph),
                                         _ptr(ray_ori), _ptr(ray_dir), _ptr(fd), _ptr(g_fd), _ptr(dist), _ptr(g_dist),
                                         _ptr(g_density), _ptr(g_sph)), "gut_backward")
        return g_density, g_sph

    def trace_bwd_unpacked(self, frame, particle_density, particle_sph, ray_ori, ray_dir, fd, g_feat, g_opa, dist, g_dist):
        """gut_backward_unpacked: upstream gradients as autograd hands them over ([H,W,3] / [H,W,1], either may be None), particle
        gradients written straight into the model's four tensor shapes — no concatenation before, no unpack pass after."""
        dev = ray_ori.device
        n = particle_density.shape[0]
        opts = dict(dtype=torch.float32, device=dev)
        g_pos, g_dns, g_rot, g_scl = torch.empty((n, 3), **opts), torch.empty((n, 1), **opts), torch.empty((n, 4), **opts), torch.empty((n, 3), **opts)
        g_sph = torch.empty_like(particle_sph, dtype=torch.float32)
        io = _abi.GutGradIO(_ptr(g_feat), _ptr(g_opa), _ptr(g_pos), _ptr(g_dns), _ptr(g_rot), _ptr(g_scl))
        _abi.check(self.lib.gut_backward_unpacked(self.handle, _stream_ptr(dev), C.byref(frame), _ptr(particle_density), _ptr(particle_sph),
                                                  _ptr(ray_ori), _ptr(ray_dir), _ptr(fd), _ptr(dist), _ptr(g_dist), C.byref(io), _ptr(g_sph)),
                   "gut_backward_unpacked")
        return g_pos, g_dns, g_rot, g_scl, g_sph

    def trace_bwd_factored(self, frame, particle_density, particle_sph, ray_ori, ray_dir, fd, g_fd, dist, g_dist):
        """gut_backward_factored: (packed gradient [N,12], view factor [N+1,3]) — see 3dgrut_amd/dp.py."""
        dev = ray_ori.device
        g_density = torch.empty_like(particle_density)
        g_radiance = torch.empty((particle_density.shape[0] + 1, 3), dtype=torch.float32, device=dev)
        _abi.check(self.lib.gut_backward_factored(self.handle, _stream_ptr(dev), C.byref(frame), _ptr(particle_density), _ptr(particle_sph),
                                                  _ptr(ray_ori), _ptr(ray_dir), _ptr(fd), _ptr(g_fd), _ptr(dist), _ptr(g_dist),
                                                  _ptr(g_density), _ptr(g_radiance)), "gut_backward_factored")
        return g_density, g_radiance

    def trace_bwd_factored_chunked(self, frame, particle_density, particle_sph, ray_ori, ray_dir, fd, g_fd, dist, g_dist, num_chunks, on_chunk):
        """gut_backward_factored_chunked: as trace_bwd_factored, but the finalisation runs in `num_chunks` particle ranges and
        on_chunk(chunk, first, count, g_density, g_radiance) is called after each range's kernels are enqueued (rows [first, first + count)
        of both tensors are complete in stream order)."""
        dev = ray_ori.device
        g_density = torch.empty_like(particle_density)
        g_radiance = torch.empty((particle_density.shape[0] + 1, 3), dtype=torch.float32, device=dev)
        errors = []

        def _cb(_user, chunk, first, count):
            try:
                on_chunk(int(chunk), int(first), int(count), g_density, g_radiance)
            except BaseException as e:   # an exception must not unwind through the C frames
                errors.append(e)
        cb = _abi.CHUNK_FN(_cb)
        rc = self.lib.gut_backward_factored_chunked(self.handle, _stream_ptr(dev), C.byref(frame), _ptr(particle_density), _ptr(particle_sph),
                                                    _ptr(ray_ori), _ptr(ray_dir), _ptr(fd), _ptr(g_fd), _ptr(dist), _ptr(g_dist),
                                                    _ptr(g_density), _ptr(g_radiance), int(num_chunks), cb, None)
        if errors:
            raise errors[0]
        _abi.check(rc, "gut_backward_factored_chunked")
        return g_density, g_radiance

    def collect_times(self):
        if not self.cfg.enable_kernel_timings:
            return {}
        f, b = C.c_float(-1), C.c_float(-1)
        _abi.check(self.lib.gut_timings(self.handle, C.byref(f), C.byref(b)), "gut_timings")
        if f.value >= 0:
            self._timings["forward_render"] = f.value
        if b.value >= 0:
            self._timings["backward_render"] = b.value
        return dict(self._timings)

    def trim(self):
        """gut_trim: hand all scratch back to the allocator (torch's pool by default); the next frame allocates afresh."""
        _abi.check(self.lib.gut_trim(self.handle), "gut_trim")

    def stats(self) -> _abi.GutStats:
        s = _abi.GutStats()
        _abi.check(self.lib.gut_stats(self.handle, C.byref(s)), "gut_stats")
        return s


class _NhtAutograd(torch.autograd.Function):
    """model.feature_type = nht: the op of threedgut_tracer/tracer.py:166-300 with per-ray features ([H,W,ray_dim] + opacity)."""

    @staticmethod
    def forward(ctx, native, frame, ray_ori, ray_dir, mog_pos, mog_rot, mog_scl, mog_dns, mog_feat, exchange=None):
        particle_density = _abi.pack_particles(mog_pos, mog_dns, mog_rot, mog_scl)
        fd, dist, cnt, vis, feats_k = native.trace_nht(frame, particle_density, mog_feat.contiguous(), ray_ori, ray_dir)
        ctx.save_for_backward(ray_ori, ray_dir, fd, dist, particle_density, feats_k)
        ctx.native, ctx.frame, ctx.exchange = native, frame, exchange
        nr = native.ray_feature_dim
        f32 = fd.float()   # always fp32 to the caller; the (possibly half) image stays in the context for the backward
        ctx.mark_non_differentiable(cnt, vis)
        ctx.set_materialize_grads(False)
        return f32[..., :nr].unsqueeze(0).contiguous(), f32[..., nr:].unsqueeze(0).contiguous(), dist, cnt, vis

    @staticmethod
    def backward(ctx, g_feat, g_opa, g_dist, _g_cnt, _g_vis):
        ray_ori, ray_dir, fd, dist, particle_density, feats_k = ctx.saved_tensors
        H, W, nr = fd.shape[0], fd.shape[1], fd.shape[2] - 1
        c = ctx.native.cfg
        if (ctx.exchange is None and c.particle_feature_dim == 48 and c.interp_point_feature_dim == 12 and c.feature_interpolation_support == 1
                and c.feature_activation_type == 2 and c.feature_activation_num_frequencies == 1 and int(c.k_buffer_size) == 0
                and not os.environ.get("GRUT_NHT_GENERIC")):
            # the default feature model runs on the pixel-pair sweeps, which take the upstream gradients as autograd delivers them and write
            # the model's four gradient tensors directly: no concatenation before, no unpack pass after
            g_pos, g_dns, g_rot, g_scl, g_features = ctx.native.trace_bwd_unpacked(
                ctx.frame, particle_density, feats_k, ray_ori, ray_dir, fd,
                None if g_feat is None else g_feat.reshape(H, W, nr).float().contiguous(), None if g_opa is None else g_opa.reshape(H, W, 1).float().contiguous(),
                dist, None if g_dist is None else g_dist.contiguous())
            return None, None, None, None, g_pos, g_rot, g_scl, g_dns, g_features, None
        g_feat = fd.new_zeros((H, W, nr), dtype=torch.float32) if g_feat is None else g_feat.reshape(H, W, nr).float()
        g_opa = fd.new_zeros((H, W, 1), dtype=torch.float32) if g_opa is None else g_opa.reshape(H, W, 1).float()
        g_fd = torch.cat([g_feat, g_opa], dim=-1).contiguous()
        g_density, g_features = ctx.native.trace_bwd_nht(ctx.frame, particle_density, feats_k, ray_ori, ray_dir, fd, g_fd, dist,
                                                         None if g_dist is None else g_dist.contiguous())
        if ctx.exchange is not None:
            # view-sharded data parallelism: the feature-row gradient has no per-view factorisation (it sums hit-dependent barycentric
            # weights), so both tensors are reduced densely, in place, before anything is unpacked
            g_density, g_features = ctx.exchange.reduce_dense(g_density, g_features)
        g_pos, g_dns, g_rot, g_scl = _abi.unpack_particle_grads(g_density)
        return None, None, None, None, g_pos, g_rot, g_scl, g_dns, g_features, None


class Tracer:
    class _Autograd(torch.autograd.Function):
        @staticmethod
        def forward(ctx, native, frame, ray_ori, ray_dir, mog_pos, mog_rot, mog_scl, mog_dns, mog_sph, raw=False, exchange=None):
            # raw: mog_rot / mog_scl / mog_dns are the model's RAW parameters, activated inside the packing kernel
            if raw:
                particle_density = _abi.activate_pack(mog_pos, mog_dns, mog_rot, mog_scl)
            else:
                particle_density = _abi.pack_particles(mog_pos, mog_dns, mog_rot, mog_scl)  # [N,12] rows, one pass (tracer.py's torch.cat)
            ctx.raw = (mog_dns, mog_rot, mog_scl) if raw else None
            particle_features = mog_sph.contiguous()
            if native.cfg.particle_feature_half and particle_features.dtype != torch.float16:
                particle_features = particle_features.to(torch.float16)   # per call, like splatRaster.cpp:90-98
            fd, dist, cnt, vis, feat, opa = native.trace(frame, particle_density, particle_features, ray_ori, ray_dir)
            ctx.save_for_backward(ray_ori, ray_dir, fd, dist, particle_density, particle_features)
            ctx.native, ctx.frame, ctx.exchange = native, frame, exchange
            # the op hands out features and opacity as separate CONTIGUOUS tensors (what render() returns, tracer.py:334-337), so
            # that autograd does not have to route their gradients back through slice / contiguous nodes and callers may
            # .view() or modify them in place
            feat = feat.unsqueeze(0)
            opa = opa.unsqueeze(0)
            ctx.mark_non_differentiable(cnt, vis)
            ctx.set_materialize_grads(False)
            return feat, opa, dist, cnt, vis

        @staticmethod
        def backward(ctx, g_feat, g_opa, g_dist, _g_cnt, _g_vis):
            ray_ori, ray_dir, fd, dist, particle_density, particle_features = ctx.saved_tensors
            H, W = fd.shape[0], fd.shape[1]
            # g_dist is None when the loss never touched pred_dist: the library then runs the variant without
            # hit-distance terms (autograd materialises zeros unless told otherwise, see set_materialize_grads)
            g_dist = None if g_dist is None else g_dist.contiguous()
            if ctx.exchange is None and ctx.raw is None and int(ctx.native.cfg.k_buffer_size) == 0:
                # the usual training path: gradients cross the boundary in autograd's own shapes (the reference concatenates the two
                # upstream gradients and slices the packed result apart again, tracer.py:226-285: two passes saved per iteration)
                g_pos, g_dns, g_rot, g_scl, g_sph = ctx.native.trace_bwd_unpacked(
                    ctx.frame, particle_density, particle_features, ray_ori, ray_dir, fd,
                    None if g_feat is None else g_feat.reshape(H, W, 3).contiguous(), None if g_opa is None else g_opa.reshape(H, W, 1).contiguous(),
                    dist, g_dist)
                return None, None, None, None, g_pos, g_rot, g_scl, g_dns, g_sph, None, None
            g_feat = fd.new_zeros((H, W, 3)) if g_feat is None else g_feat.reshape(H, W, 3)
            g_opa = fd.new_zeros((H, W, 1)) if g_opa is None else g_opa.reshape(H, W, 1)
            g_fd = torch.cat([g_feat, g_opa], dim=-1)
            if ctx.exchange is None:
                g_density, g_sph = ctx.native.trace_bwd(ctx.frame, particle_density, particle_features, ray_ori, ray_dir, fd, g_fd, dist, g_dist)
            else:
                # view-sharded data parallelism (3dgrut_amd/dp.py): the packed gradient is all-reduced and the SH gradient is
                # rebuilt from the gathered per-view factors, before anything is unpacked
                if getattr(ctx.exchange, "chunks", 1) > 1 and int(ctx.native.cfg.k_buffer_size) == 0:
                    # pipelined: the collectives of particle range i are issued while the finalisation kernels of range i + 1 run
                    g_density, g_sph = ctx.exchange.reduce_packed_pipelined(
                        lambda n, cb: ctx.native.trace_bwd_factored_chunked(ctx.frame, particle_density, particle_features, ray_ori, ray_dir, fd, g_fd,
                                                                            dist, g_dist, n, cb),
                        particle_density, int(ctx.frame.n_active_features), int(ctx.native.cfg.particle_radiance_sph_degree))
                else:
                    g_density, g_radiance = ctx.native.trace_bwd_factored(ctx.frame, particle_density, particle_features, ray_ori, ray_dir,
                                                                          fd, g_fd, dist, g_dist)
                    g_density, g_sph = ctx.exchange.reduce_packed(g_density, g_radiance, particle_density, int(ctx.frame.n_active_features),
                                                                  int(ctx.native.cfg.particle_radiance_sph_degree))
            # views into the packed gradient, as the reference returns them (tracer.py:268-285): no copies
            if ctx.raw is not None:   # chain rule to the raw parameters, four contiguous tensors in one pass
                g_pos, g_dns, g_rot, g_scl = _abi.activate_pack_backward(*ctx.raw, g_density)
                return None, None, None, None, g_pos, g_rot, g_scl, g_dns, g_sph, None, None
            # the reference returns strided slices of the packed gradient (tracer.py:268-285) and autograd clones each of
            # them when it accumulates; one pass writes the four tensors contiguously instead
            g_pos, g_dns, g_rot, g_scl = _abi.unpack_particle_grads(g_density)
            return None, None, None, None, g_pos, g_rot, g_scl, g_dns, g_sph, None, None

    def __init__(self, conf):
        self.device = "cuda"
        self.conf = conf
        self.tracer_wrapper = _GutNative(gut_config_from_conf(conf))   # raises without a ROCm GPU: there is no CPU fallback
        self._fused_activations = fused_activations_requested(conf)
        # set to a 3dgrut_amd.dp.FactoredGradientExchange to have the backward exchange its gradients across ranks (one view
        # per GPU); None (default) = single-GPU behaviour, exactly the reference's
        self.gradient_exchange = None

    @property
    def timings(self):
        return self.tracer_wrapper.collect_times()

    def build_acc(self, gaussians, rebuild=True):
        pass  # no-op for 3DGUT (tracer.py:301-302)

    def _constant_normals(self, like):
        """normalize(ones) (tracer.py:345): a constant per resolution — built once instead of four elementwise passes per frame."""
        key = (tuple(like.shape), like.device, like.dtype)
        if getattr(self, "_normals_key", None) != key:
            self._normals = torch.nn.functional.normalize(torch.ones_like(like), dim=3)
            self._normals_key = key
        return self._normals

    def render(self, gaussians, gpu_batch, train=False, frame_id=0):
        rays_o = gpu_batch.rays_ori if not isinstance(gpu_batch, dict) else gpu_batch["rays_ori"]
        rays_d = gpu_batch.rays_dir if not isinstance(gpu_batch, dict) else gpu_batch["rays_dir"]
        T0 = gpu_batch["T_to_world"] if isinstance(gpu_batch, dict) else gpu_batch.T_to_world
        T1 = (gpu_batch.get("T_to_world_end") if isinstance(gpu_batch, dict) else getattr(gpu_batch, "T_to_world_end", None))
        in_world = bool(gpu_batch.get("rays_in_world_space", False) if isinstance(gpu_batch, dict) else getattr(gpu_batch, "rays_in_world_space", False))
        dev_poses = torch.is_tensor(T0) and T0.is_cuda and not in_world and (T1 is None or (torch.is_tensor(T1) and T1.is_cuda))
        cam, pose_start, pose_end = camera_from_batch(gpu_batch, poses_on_device=dev_poses)
        H, W = int(rays_o.shape[1]), int(rays_o.shape[2])
        native = self.tracer_wrapper
        frame = native.make_frame(frame_id, gaussians.n_active_features, gaussians.num_gaussians, H, W, cam, pose_start, pose_end)
        if dev_poses:  # the library derives the sensor poses on the GPU: no host round trip, no stream drain
            t0 = T0.detach().reshape(-1, 4, 4)[0].to(torch.float32).contiguous()
            t1 = None if T1 is None else T1.detach().reshape(-1, 4, 4)[0].to(torch.float32).contiguous()
            frame.device_T_to_world = t0.data_ptr()
            frame.device_T_to_world_end = None if t1 is None else t1.data_ptr()
            frame._keepalive = (t0, t1)
        feats = gaussians.get_features()
        if native.cfg.feature_transform_type:   # neural harmonic features (K = 0)
            if feats.shape[1] != native.cfg.particle_feature_dim:
                raise ValueError(f"features have {feats.shape[1]} columns, expected nht_features.dim = {native.cfg.particle_feature_dim}")
            pred_features, pred_opacity, pred_dist, hits_count, mog_visibility = _NhtAutograd.apply(
                native, frame, rays_o.contiguous().float(), rays_d.contiguous().float(), gaussians.positions.contiguous(),
                gaussians.get_rotation().contiguous(), gaussians.get_scale().contiguous(), gaussians.get_density().contiguous(), feats.contiguous(),
                self.gradient_exchange)
            timings = native.collect_times()
            return {"pred_features": pred_features, "pred_opacity": pred_opacity, "pred_dist": pred_dist.unsqueeze(0),
                    "pred_normals": torch.nn.functional.normalize(torch.ones_like(pred_features), dim=3), "hits_count": hits_count.unsqueeze(0),
                    "frame_time_ms": timings["forward_render"] if "forward_render" in timings else 0.0, "mog_visibility": mog_visibility}
        if feats.shape[1] != 3 * native.ncoef:
            raise ValueError(f"features have {feats.shape[1]} columns, expected {3 * native.ncoef} for SH degree "
                             f"{native.cfg.particle_radiance_sph_degree}")
        if self._fused_activations and has_standard_activations(gaussians):
            pred_features, pred_opacity, pred_dist, hits_count, mog_visibility = Tracer._Autograd.apply(
                native, frame, rays_o.contiguous().float(), rays_d.contiguous().float(),
                gaussians.positions.contiguous(), gaussians.rotation.contiguous(), gaussians.scale.contiguous(),
                gaussians.density.contiguous(), feats.contiguous(), True, self.gradient_exchange)
        else:
            pred_features, pred_opacity, pred_dist, hits_count, mog_visibility = Tracer._Autograd.apply(
                native, frame, rays_o.contiguous().float(), rays_d.contiguous().float(),
                gaussians.positions.contiguous(), gaussians.get_rotation().contiguous(), gaussians.get_scale().contiguous(),
                gaussians.get_density().contiguous(), feats.contiguous(), False, self.gradient_exchange)
        if getattr(gaussians, "ray_feature_dim", 3) != 3:
            raise NotImplementedError("3dgrut_amd: only SH radiance features (ray_feature_dim = 3) are supported")
        timings = native.collect_times()
        return {
            "pred_features": pred_features,
            "pred_opacity": pred_opacity,
            "pred_dist": pred_dist.unsqueeze(0),
            "pred_normals": self._constant_normals(pred_features),
            "hits_count": hits_count.unsqueeze(0),
            "frame_time_ms": timings["forward_render"] if "forward_render" in timings else 0.0,
            "mog_visibility": mog_visibility,
        }

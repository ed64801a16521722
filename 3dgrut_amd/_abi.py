"""ctypes mirror of include/grut_amd.h and the loader of the HIP shared library.

The library is the product: there is no CPU fallback.  `load_library()` raises if
`libgrut_amd.so` has not been built (run `python -c "import __graft_entry__ as g; g.build()"`
or `python -m 3dgrut_amd.build` equivalent, see build.py).
"""
from __future__ import annotations

import ctypes as C
import os

ABI_VERSION = 5
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgrut_amd.so")

# ---- enums (include/grut_amd.h) -------------------------------------------------
SHUTTER_ROLLING_TOP_TO_BOTTOM, SHUTTER_ROLLING_LEFT_TO_RIGHT, SHUTTER_ROLLING_BOTTOM_TO_TOP, \
    SHUTTER_ROLLING_RIGHT_TO_LEFT, SHUTTER_GLOBAL = range(5)
CAMERA_OPENCV_PINHOLE, CAMERA_OPENCV_FISHEYE, CAMERA_FTHETA = range(3)
FTHETA_PIXELDIST_TO_ANGLE, FTHETA_ANGLE_TO_PIXELDIST = range(2)

GUT_STAGES = ["project", "depth_sort", "scan", "expand", "tile_sort", "tile_ranges", "render_fwd", "render_bwd", "project_bwd"]

GRUT_OK = 0
STATUS_NAMES = {0: "OK", -1: "BAD_INPUT", -2: "RUNTIME", -3: "NOT_READY", -4: "UNSUPPORTED"}


class GrutCamera(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("shutter", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
        ("principal_point", C.c_float * 2), ("focal_length", C.c_float * 2),
        ("radial", C.c_float * 6), ("tangential", C.c_float * 2), ("thin_prism", C.c_float * 4),
        ("max_angle", C.c_float), ("ftheta_reference_poly", C.c_int32),
        ("ftheta_pixeldist_to_angle", C.c_float * 6), ("ftheta_angle_to_pixeldist", C.c_float * 6),
        ("ftheta_linear_cde", C.c_float * 3),
    ]


class GutConfig(C.Structure):
    _fields_ = [
        ("particle_kernel_degree", C.c_int32), ("particle_kernel_min_response", C.c_float),
        ("particle_kernel_min_alpha", C.c_float), ("particle_kernel_max_alpha", C.c_float),
        ("min_transmittance", C.c_float), ("particle_radiance_sph_degree", C.c_int32),
        ("enable_hitcounts", C.c_int32), ("enable_kernel_timings", C.c_int32),
        ("ut_alpha", C.c_float), ("ut_beta", C.c_float), ("ut_kappa", C.c_float),
        ("ut_in_image_margin_factor", C.c_float), ("ut_require_all_sigma_points_valid", C.c_int32),
        ("n_rolling_shutter_iterations", C.c_int32), ("k_buffer_size", C.c_int32),
        ("global_z_order", C.c_int32), ("rect_bounding", C.c_int32),
        ("tight_opacity_bounding", C.c_int32), ("tile_based_culling", C.c_int32),
        ("particle_feature_half", C.c_int32), ("feature_output_half", C.c_int32),
        ("feature_transform_type", C.c_int32), ("particle_feature_dim", C.c_int32), ("interp_point_feature_dim", C.c_int32),
        ("feature_interpolation_support", C.c_int32), ("feature_activation_type", C.c_int32), ("feature_activation_num_frequencies", C.c_int32),
    ]


class GutFrame(C.Structure):
    _fields_ = [
        ("frame_id", C.c_uint32), ("n_active_features", C.c_int32), ("num_particles", C.c_uint32),
        ("width", C.c_int32), ("height", C.c_int32), ("camera", GrutCamera),
        ("pose_start", C.c_float * 7), ("pose_end", C.c_float * 7),
        ("device_T_to_world", C.c_void_p), ("device_T_to_world_end", C.c_void_p), ("out_features", C.c_void_p), ("out_opacity", C.c_void_p),
    ]


class GutStats(C.Structure):
    _fields_ = [
        ("num_particles", C.c_uint32), ("num_visible", C.c_uint32), ("num_intersections", C.c_uint64),
        ("num_tiles", C.c_uint32), ("key_bits", C.c_uint32),
        ("fwd_entries_evaluated", C.c_uint64), ("fwd_entries_accepted", C.c_uint64),
        ("bwd_entries_evaluated", C.c_uint64), ("bwd_entries_accepted", C.c_uint64),
    ]


class GrtConfig(C.Structure):
    _fields_ = [
        ("particle_kernel_degree", C.c_int32), ("particle_kernel_min_response", C.c_float),
        ("particle_kernel_min_alpha", C.c_float), ("particle_kernel_max_alpha", C.c_float),
        ("particle_kernel_density_clamping", C.c_int32), ("particle_radiance_sph_degree", C.c_int32),
        ("enable_normals", C.c_int32), ("enable_hitcounts", C.c_int32), ("enable_kernel_timings", C.c_int32),
        ("max_hits_per_trace", C.c_int32), ("particle_feature_half", C.c_int32), ("feature_output_half", C.c_int32),
        ("feature_transform_type", C.c_int32), ("particle_feature_dim", C.c_int32), ("interp_point_feature_dim", C.c_int32),
        ("feature_interpolation_support", C.c_int32), ("feature_activation_type", C.c_int32), ("feature_activation_num_frequencies", C.c_int32),
        ("primitive_type", C.c_int32), ("pipeline_type", C.c_int32),
    ]


class GrtFrame(C.Structure):
    _fields_ = [
        ("frame_id", C.c_uint32), ("sph_degree", C.c_int32), ("min_transmittance", C.c_float),
        ("num_particles", C.c_uint32), ("width", C.c_int32), ("height", C.c_int32),
        ("ray_to_world", C.c_float * 12), ("keep_hits_for_backward", C.c_int32), ("device_ray_to_world", C.c_void_p),
    ]


class GrtStats(C.Structure):
    _fields_ = [
        ("num_particles", C.c_uint32), ("num_nodes", C.c_uint32), ("nodes_visited", C.c_uint64),
        ("candidates", C.c_uint64), ("processed_hits", C.c_uint64), ("scene_aabb", C.c_float * 6),
        ("list_entries", C.c_uint64), ("packet_tests", C.c_uint64), ("list_batches", C.c_uint64),
        ("bwd_rederived_rays", C.c_uint32), ("bwd_premise_rays", C.c_uint32),
        ("bwd_atomic_instructions", C.c_uint64), ("bwd_atomic_words", C.c_uint64),
    ]


class GrutAdamGroup(C.Structure):
    _fields_ = [
        ("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
        ("row_width", C.c_uint32), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
    ]


VIS_NONE, VIS_BOOL_U8, VIS_INT32, VIS_FLOAT_BITS = range(4)
# GrtConfig::primitive_type (render.primitive_type, optixTracer.cpp:176-201)
GRT_PRIMITIVES = {"instances": 0, "icosahedron": 1, "octahedron": 2, "tetrahedron": 3, "diamond": 4, "custom": 5, "trisurfel": 6, "trihexa": 7, "sphere": 8}


class GrtTexture(C.Structure):
    """include/grut_amd.h: GrtTexture — [height, width, channels] f32 in device memory (data = None: no texture)."""
    _fields_ = [("data", C.c_void_p), ("height", C.c_int32), ("width", C.c_int32), ("channels", C.c_int32)]


class GrtMaterial(C.Structure):
    """include/grut_amd.h: GrtMaterial (PBRMaterial of the reference's playground)."""
    _fields_ = [("diffuse", GrtTexture), ("emissive", GrtTexture), ("metallic_roughness", GrtTexture), ("normal", GrtTexture),
                ("diffuse_factor", C.c_float * 4), ("emissive_factor", C.c_float * 3), ("metallic_factor", C.c_float), ("roughness_factor", C.c_float),
                ("transmission_factor", C.c_float), ("ior", C.c_float), ("alpha_cutoff", C.c_float), ("alpha_mode", C.c_uint32)]


class GrtMesh(C.Structure):
    """include/grut_amd.h: GrtMesh (device pointers of the hybrid tracer's triangle mesh; the material table is a host array)."""
    _fields_ = [("num_vertices", C.c_uint32), ("num_faces", C.c_uint32), ("vertices", C.c_void_p), ("triangles", C.c_void_p),
                ("vertex_normals", C.c_void_p), ("vertex_tangents", C.c_void_p), ("vertex_has_tangents", C.c_void_p), ("prim_type", C.c_void_p),
                ("mat_uv", C.c_void_p), ("mat_id", C.c_void_p), ("refractive_index", C.c_void_p), ("num_materials", C.c_uint32),
                ("materials", C.POINTER(GrtMaterial)), ("envmap", GrtTexture), ("envmap_offset", C.c_float * 2)]


class GrtHybridOptions(C.Structure):
    _fields_ = [("playground_opts", C.c_uint32), ("max_pbr_bounces", C.c_uint32), ("frame_number", C.c_uint32)]



class GutGradIO(C.Structure):
    """include/grut_amd.h: GutGradIO (gradient tensors of gut_backward_unpacked in the caller's own layout)."""
    _fields_ = [("grad_features", C.c_void_p), ("grad_opacity", C.c_void_p), ("grad_positions", C.c_void_p), ("grad_density", C.c_void_p),
                ("grad_rotation", C.c_void_p), ("grad_scale", C.c_void_p)]

# every symbol include/grut_amd.h declares (checked by tests/test_abi.py)
EXPORTED_SYMBOLS = [
    "gut_create", "gut_destroy", "gut_forward", "gut_backward", "gut_backward_unpacked", "gut_backward_factored", "gut_backward_factored_chunked", "grut_sph_grad_from_views", "gut_timings", "gut_stats",
    "gut_profile_enable", "gut_profile_select", "gut_profile_read",
    "gut_debug_fetch", "gut_debug_fetch_work", "grut_debug_pose_from_c2w", "grut_debug_frame_poses", "grut_sort_pairs_u32", "grut_sort_scratch_bytes", "grut_inclusive_scan_u32",
    "grut_scan_scratch_bytes",
    "grt_create", "grt_destroy", "grt_build_bvh", "grt_forward", "grt_backward", "grt_timings", "grt_stats", "grt_debug_fetch_work",
    "grt_debug_forward_hits", "grt_debug_fetch_instances", "grt_debug_fetch_custom_boxes", "grt_debug_fetch_lists", "grt_debug_backward_signature", "grt_build_mesh_bvh", "grt_trace_hybrid",
    "grut_selective_adam_update", "grut_pack_particles", "grut_unpack_particle_grads", "grut_activate_pack", "grut_activate_pack_backward",
    "grut_last_error", "grut_abi_version", "grut_set_allocator", "gut_trim", "grt_trim",
]

_lib = None
ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint64)   # GrutAllocFn / GrutFreeFn of include/grut_amd.h
FREE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
CHUNK_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32)   # GrutChunkFn: (user, chunk, first particle, particles)


def _declare(lib):
    vp, fp, ip, up = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p  # device pointers travel as integers
    lib.gut_create.argtypes = [C.POINTER(GutConfig), C.POINTER(C.c_void_p)]
    lib.gut_create.restype = C.c_int
    lib.gut_destroy.argtypes = [C.c_void_p]
    lib.gut_destroy.restype = None
    lib.gut_forward.argtypes = [C.c_void_p, vp, C.POINTER(GutFrame), fp, fp, fp, fp, fp, fp, fp, ip]
    lib.gut_forward.restype = C.c_int
    lib.gut_backward.argtypes = [C.c_void_p, vp, C.POINTER(GutFrame)] + [fp] * 10
    lib.gut_backward.restype = C.c_int
    lib.gut_backward_factored.argtypes = [C.c_void_p, vp, C.POINTER(GutFrame)] + [fp] * 10
    lib.gut_backward_factored.restype = C.c_int
    lib.gut_backward_factored_chunked.argtypes = [C.c_void_p, vp, C.POINTER(GutFrame)] + [fp] * 10 + [C.c_uint32, CHUNK_FN, C.c_void_p]
    lib.gut_backward_factored_chunked.restype = C.c_int
    lib.gut_backward_unpacked.argtypes = [C.c_void_p, vp, C.POINTER(GutFrame)] + [fp] * 7 + [C.POINTER(GutGradIO), fp]
    lib.gut_backward_unpacked.restype = C.c_int
    lib.grut_sph_grad_from_views.argtypes = [vp, C.c_uint32, C.c_uint32, fp, fp, C.c_uint32, C.c_int32, C.c_int32, C.c_float, fp]
    lib.grut_sph_grad_from_views.restype = C.c_int
    lib.gut_timings.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.gut_timings.restype = C.c_int
    lib.gut_stats.argtypes = [C.c_void_p, C.POINTER(GutStats)]
    lib.gut_stats.restype = C.c_int
    lib.gut_profile_enable.argtypes = [C.c_void_p, C.c_int]
    lib.gut_profile_enable.restype = C.c_int
    lib.gut_profile_select.argtypes = [C.c_void_p, C.c_uint32]
    lib.gut_profile_select.restype = C.c_int
    lib.gut_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.gut_profile_read.restype = C.c_int
    lib.gut_debug_fetch_work.argtypes = [C.c_void_p, vp, up, C.c_uint64]
    lib.gut_debug_fetch_work.restype = C.c_int
    lib.grut_debug_pose_from_c2w.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.grut_debug_pose_from_c2w.restype = C.c_int
    lib.grut_debug_frame_poses.argtypes = [vp, C.c_int, vp, vp, C.POINTER(C.c_float)]
    lib.grut_debug_frame_poses.restype = C.c_int
    lib.gut_debug_fetch.argtypes = [C.c_void_p, vp] + [up] * 8
    lib.gut_debug_fetch.restype = C.c_int
    lib.grut_sort_pairs_u32.argtypes = [vp, C.c_uint32, C.c_int, C.c_int, up, up, up, up, vp, C.c_uint64,
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    lib.grut_sort_pairs_u32.restype = C.c_int
    lib.grut_sort_scratch_bytes.argtypes = [C.c_uint32]
    lib.grut_sort_scratch_bytes.restype = C.c_uint64
    lib.grut_inclusive_scan_u32.argtypes = [vp, C.c_uint32, up, up, vp, C.c_uint64]
    lib.grut_inclusive_scan_u32.restype = C.c_int
    lib.grut_scan_scratch_bytes.argtypes = [C.c_uint32]
    lib.grut_scan_scratch_bytes.restype = C.c_uint64
    lib.grt_debug_fetch_lists.argtypes = [C.c_void_p, vp, up, up, C.c_uint64]
    lib.grt_debug_fetch_lists.restype = C.c_int
    lib.grt_create.argtypes = [C.POINTER(GrtConfig), C.POINTER(C.c_void_p)]
    lib.grt_create.restype = C.c_int
    lib.grt_destroy.argtypes = [C.c_void_p]
    lib.grt_destroy.restype = None
    lib.grt_build_bvh.argtypes = [C.c_void_p, vp, C.c_uint32, fp, fp, fp, fp, C.c_int, C.c_int]
    lib.grt_build_bvh.restype = C.c_int
    lib.grt_forward.argtypes = [C.c_void_p, vp, C.POINTER(GrtFrame)] + [fp] * 9 + [ip]
    lib.grt_forward.restype = C.c_int
    lib.grt_backward.argtypes = [C.c_void_p, vp, C.POINTER(GrtFrame)] + [fp] * 14
    lib.grt_backward.restype = C.c_int
    lib.grt_debug_forward_hits.argtypes = [C.c_void_p, vp, C.POINTER(GrtFrame)] + [fp] * 9 + [ip, up, up, C.c_uint32]
    lib.grt_debug_forward_hits.restype = C.c_int
    lib.grt_debug_fetch_instances.argtypes = [C.c_void_p, vp, fp]
    lib.grt_debug_fetch_instances.restype = C.c_int
    lib.grt_debug_fetch_custom_boxes.argtypes = [C.c_void_p, vp, fp]
    lib.grt_debug_fetch_custom_boxes.restype = C.c_int
    lib.grt_debug_backward_signature.argtypes = [C.c_void_p, up, up]
    lib.grt_debug_backward_signature.restype = C.c_int
    lib.grt_build_mesh_bvh.argtypes = [C.c_void_p, vp, C.c_uint32, fp, C.c_uint32, ip, C.c_int, C.c_int]
    lib.grt_build_mesh_bvh.restype = C.c_int
    lib.grt_trace_hybrid.argtypes = [C.c_void_p, vp, C.POINTER(GrtFrame), fp, fp, fp, fp, fp, C.POINTER(GrtMesh), C.POINTER(GrtHybridOptions), fp, fp, fp, up]
    lib.grt_trace_hybrid.restype = C.c_int
    lib.grt_timings.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.grt_timings.restype = C.c_int
    lib.grut_set_allocator.argtypes = [ALLOC_FN, FREE_FN, C.c_void_p]
    lib.grut_set_allocator.restype = C.c_int
    lib.gut_trim.argtypes = [C.c_void_p]
    lib.gut_trim.restype = C.c_int
    lib.grt_trim.argtypes = [C.c_void_p]
    lib.grt_trim.restype = C.c_int
    lib.grt_debug_fetch_work.argtypes = [C.c_void_p, vp, up, C.c_uint64]
    lib.grt_debug_fetch_work.restype = C.c_int
    lib.grt_stats.argtypes = [C.c_void_p, C.POINTER(GrtStats)]
    lib.grt_stats.restype = C.c_int
    lib.grut_selective_adam_update.argtypes = [vp, C.POINTER(GrutAdamGroup), C.c_int, C.c_uint32, vp, C.c_int]
    lib.grut_selective_adam_update.restype = C.c_int
    lib.grut_pack_particles.argtypes = [vp, C.c_uint32, fp, fp, fp, fp, fp]
    lib.grut_pack_particles.restype = C.c_int
    lib.grut_unpack_particle_grads.argtypes = [vp, C.c_uint32] + [fp] * 5
    lib.grut_unpack_particle_grads.restype = C.c_int
    lib.grut_activate_pack.argtypes = [vp, C.c_uint32, fp, fp, fp, fp, fp]
    lib.grut_activate_pack.restype = C.c_int
    lib.grut_activate_pack_backward.argtypes = [vp, C.c_uint32] + [fp] * 8
    lib.grut_activate_pack_backward.restype = C.c_int
    lib.grut_last_error.argtypes = []
    lib.grut_last_error.restype = C.c_char_p
    lib.grut_abi_version.argtypes = []
    lib.grut_abi_version.restype = C.c_int


def load_library(path: str | None = None):
    """Load libgrut_amd.so (built in-tree by build.py).  Fails loudly if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("GRUT_AMD_LIB", LIB_PATH)
    # PyTorch-ROCm bundles its own libamdhip64 / libhsa-runtime64; it must be resident first so that this library
    # binds to the same HIP runtime (same device context, streams and allocations) instead of a second copy.
    import torch  # noqa: F401
    if not os.path.exists(p):
        raise RuntimeError(
            f"3dgrut_amd: HIP library not found at {p}. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc, --offload-arch=gfx950). "
            "There is no CPU fallback.")
    lib = C.CDLL(p)
    _declare(lib)
    ver = lib.grut_abi_version()
    if ver != ABI_VERSION:
        raise RuntimeError(f"3dgrut_amd: ABI mismatch, library {ver} vs python {ABI_VERSION}; rebuild")
    if path is None:
        _lib = lib
        if torch.cuda.is_available() and not os.environ.get("GRUT_AMD_HIP_MALLOC"):
            use_torch_allocator(lib)
    return lib


# ---- scratch through the caller's allocator (grut_set_allocator) ----------------------------------------------------------------
allocator_stats = {"allocs": 0, "frees": 0, "live_bytes": 0}   # (tests: no allocator traffic in steady state)
_allocator_keepalive = {}


def use_torch_allocator(lib=None, enable=True):
    """Route the library's grow-only scratch through torch's caching allocator (blocks belong to the stream that is current when the
    library asks, which is the stream the plugins pass to every call): a grown buffer's old block returns to the pool the caller's
    tensors come from, and no hipMalloc / hipFree — each an implicit device synchronisation — happens once the pool is warm.
    GRUT_AMD_HIP_MALLOC=1 (or enable=False) keeps hipMalloc / hipFree."""
    import atexit
    import torch
    lib = lib or load_library()
    if not enable:
        check(lib.grut_set_allocator(ALLOC_FN(0), FREE_FN(0), None), "grut_set_allocator")
        _allocator_keepalive.clear()
        return
    sizes = {}

    def _alloc(_user, nbytes):
        try:
            ptr = torch.cuda.caching_allocator_alloc(int(nbytes))
        except Exception:   # out of memory: the library reports the failed size
            return None
        sizes[ptr] = int(nbytes)
        allocator_stats["allocs"] += 1
        allocator_stats["live_bytes"] += int(nbytes)
        return ptr

    def _free(_user, ptr):
        try:
            allocator_stats["frees"] += 1
            allocator_stats["live_bytes"] -= sizes.pop(ptr, 0)
            torch.cuda.caching_allocator_delete(ptr)
        except Exception:
            pass

    a, f = ALLOC_FN(_alloc), FREE_FN(_free)
    check(lib.grut_set_allocator(a, f, None), "grut_set_allocator")
    if not _allocator_keepalive:   # at interpreter exit the callbacks go away first: handles destroyed later must not call them
        atexit.register(lambda: lib.grut_set_allocator(ALLOC_FN(0), FREE_FN(0), None))
    _allocator_keepalive["fns"] = (a, f, sizes)


def pack_particles(mog_pos, mog_dns, mog_rot, mog_scl):
    """[N,12] ParticleDensity rows from the four activated Gaussian tensors (CUDA fp32), one HIP pass instead of torch.cat."""
    import torch
    lib = load_library()
    n = int(mog_pos.shape[0])
    if n == 0:
        return torch.empty((0, 12), dtype=torch.float32, device=mog_pos.device)
    parts = [t.detach().reshape(n, -1).contiguous().float() for t in (mog_pos, mog_dns, mog_rot, mog_scl)]
    out = torch.empty((n, 12), dtype=torch.float32, device=mog_pos.device)
    stream = C.c_void_p(torch.cuda.current_stream(mog_pos.device).cuda_stream)
    check(lib.grut_pack_particles(stream, n, *[C.c_void_p(p.data_ptr()) for p in parts], C.c_void_p(out.data_ptr())), "grut_pack_particles")
    return out


def sph_grad_from_views(view_factors, positions, n_active_features, sph_degree, scale=1.0, out=None):
    """Sum over views of the SH-coefficient gradients from the gathered view factors of gut_backward_factored:
    view_factors [V, N+1, 3] (row N of each view = its sensor position), positions [N, 3] or packed [N, 12] rows.
    out: a contiguous [N, 3 (deg+1)^2] float32 block to write (a row range of a larger tensor: the pipelined exchange)."""
    import torch
    lib = load_library()
    view_factors = view_factors.contiguous()
    positions = positions.contiguous()
    v, n = int(view_factors.shape[0]), int(view_factors.shape[1]) - 1
    assert view_factors.shape[2] == 3 and positions.shape[0] == n and positions.shape[1] in (3, 12)
    if out is None:
        out = torch.empty((n, 3 * (sph_degree + 1) ** 2), dtype=torch.float32, device=view_factors.device)
    assert out.is_contiguous() and tuple(out.shape) == (n, 3 * (sph_degree + 1) ** 2) and out.dtype == torch.float32
    stream = C.c_void_p(torch.cuda.current_stream(view_factors.device).cuda_stream)
    check(lib.grut_sph_grad_from_views(stream, n, v, C.c_void_p(view_factors.data_ptr()), C.c_void_p(positions.data_ptr()), int(positions.shape[1]),
                                       int(n_active_features), int(sph_degree), float(scale), C.c_void_p(out.data_ptr())), "grut_sph_grad_from_views")
    return out


def unpack_particle_grads(g_packed):
    """packed [N,12] gradient -> (g_positions, g_density, g_rotation, g_scale), four contiguous tensors in one pass."""
    import torch
    lib = load_library()
    n = int(g_packed.shape[0])
    g_packed = g_packed.contiguous()
    opts = dict(dtype=torch.float32, device=g_packed.device)
    outs = [torch.empty((n, 3), **opts), torch.empty((n, 1), **opts), torch.empty((n, 4), **opts), torch.empty((n, 3), **opts)]
    stream = C.c_void_p(torch.cuda.current_stream(g_packed.device).cuda_stream)
    check(lib.grut_unpack_particle_grads(stream, n, C.c_void_p(g_packed.data_ptr()), *[C.c_void_p(o.data_ptr()) for o in outs]),
          "grut_unpack_particle_grads")
    return outs


def activate_pack(pos, raw_dns, raw_rot, raw_scl):
    """[N,12] rows from positions and the RAW density / rotation / scale (sigmoid, normalize, exp applied in the same pass)."""
    import torch
    lib = load_library()
    n = int(pos.shape[0])
    parts = [t.detach().reshape(n, -1).contiguous().float() for t in (pos, raw_dns, raw_rot, raw_scl)]
    out = torch.empty((n, 12), dtype=torch.float32, device=pos.device)
    stream = C.c_void_p(torch.cuda.current_stream(pos.device).cuda_stream)
    check(lib.grut_activate_pack(stream, n, *[C.c_void_p(p.data_ptr()) for p in parts], C.c_void_p(out.data_ptr())), "grut_activate_pack")
    return out


def activate_pack_backward(raw_dns, raw_rot, raw_scl, g_packed):
    """-> (g_positions [N,3], g_raw_density [N,1], g_raw_rotation [N,4], g_raw_scale [N,3]), contiguous."""
    import torch
    lib = load_library()
    n = int(g_packed.shape[0])
    ins = [t.detach().reshape(n, -1).contiguous().float() for t in (raw_dns, raw_rot, raw_scl)]
    g_packed = g_packed.contiguous()
    opts = dict(dtype=torch.float32, device=g_packed.device)
    outs = [torch.empty((n, 3), **opts), torch.empty((n, 1), **opts), torch.empty((n, 4), **opts), torch.empty((n, 3), **opts)]
    stream = C.c_void_p(torch.cuda.current_stream(g_packed.device).cuda_stream)
    check(lib.grut_activate_pack_backward(stream, n, *[C.c_void_p(p.data_ptr()) for p in ins], C.c_void_p(g_packed.data_ptr()),
                                          *[C.c_void_p(o.data_ptr()) for o in outs]), "grut_activate_pack_backward")
    return outs


def check(status: int, what: str):
    if status != GRUT_OK:
        lib = load_library()
        msg = lib.grut_last_error()
        raise RuntimeError(f"3dgrut_amd: {what} failed with {STATUS_NAMES.get(status, status)}: "
                           f"{msg.decode() if msg else ''}")

/*
 * grut_amd.h — C-ABI of the MI355X-native 3DGUT / 3DGRT renderer plugin.
 *
 * This is the drop-in boundary: plain pointers and sizes, no
 * torch types.  Every entry point replaces one method of the reference's
 * pybind11 classes (citations are into /root/reference):
 *
 *   gut_create / gut_destroy   <- SplatRaster(json)            threedgut_tracer/bindings.cpp:103-109,
 *                                                              src/splatRaster.cpp:163-181
 *   gut_forward                <- SplatRaster::trace            include/3dgut/splatRaster.h:47-63,
 *                                                              src/splatRaster.cpp:184-261
 *   gut_backward               <- SplatRaster::trace_bwd        include/3dgut/splatRaster.h:65-86,
 *                                                              src/splatRaster.cpp:264-350
 *   gut_timings                <- SplatRaster::collect_times    src/splatRaster.cpp:352-382
 *   grt_create / grt_destroy   <- OptixTracer(...)              threedgrt_tracer/include/3dgrt/optixTracer.h:128-147
 *   grt_build_bvh              <- OptixTracer::build_bvh        optixTracer.h:149-155, src/optixTracer.cpp:616-890
 *   grt_forward                <- OptixTracer::trace            optixTracer.h:157-166, src/optixTracer.cpp:893-960
 *   grt_backward               <- OptixTracer::trace_bwd        optixTracer.h:168-177, src/optixTracer.cpp:962-1031
 *
 * Ownership: every I/O buffer is allocated by the caller (PyTorch) on the GPU
 * the handle was created on; a handle owns only grow-only scratch (and, for
 * 3DGRT, the BVH).  All work is enqueued on the hipStream_t passed in; the
 * backward of a frame must be enqueued on the same stream as its forward
 * (same rule as gutRenderer.cu:436-440).  Functions return 0 on success and a
 * negative GrutStatus otherwise; nothing throws across this boundary.
 */
#ifndef GRUT_AMD_H
#define GRUT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (mirrors ErrorCode, 3dgut/utils/status.h:22-29) -------- */
typedef enum GrutStatus {
    GRUT_OK              = 0,
    GRUT_ERR_BAD_INPUT   = -1,
    GRUT_ERR_RUNTIME     = -2, /* a HIP call failed; see grut_last_error() */
    GRUT_ERR_NOT_READY   = -3, /* backward without a matching forward / BVH not built */
    GRUT_ERR_UNSUPPORTED = -4
} GrutStatus;

/* ---- camera model (sensors/cameraModels.h:22-72) ------------------------- */
enum { GRUT_SHUTTER_ROLLING_TOP_TO_BOTTOM = 0,
       GRUT_SHUTTER_ROLLING_LEFT_TO_RIGHT = 1,
       GRUT_SHUTTER_ROLLING_BOTTOM_TO_TOP = 2,
       GRUT_SHUTTER_ROLLING_RIGHT_TO_LEFT = 3,
       GRUT_SHUTTER_GLOBAL                = 4 };

enum { GRUT_CAMERA_OPENCV_PINHOLE = 0,
       GRUT_CAMERA_OPENCV_FISHEYE = 1,
       GRUT_CAMERA_FTHETA         = 2 };

enum { GRUT_FTHETA_PIXELDIST_TO_ANGLE = 0,
       GRUT_FTHETA_ANGLE_TO_PIXELDIST = 1 };

typedef struct GrutCamera {
    int32_t model;              /* GRUT_CAMERA_*  */
    int32_t shutter;            /* GRUT_SHUTTER_* */
    int32_t width, height;      /* resolution of the ray image */
    float principal_point[2];
    float focal_length[2];      /* pinhole + fisheye */
    float radial[6];            /* pinhole: k1..k6; fisheye: k1..k4 */
    float tangential[2];        /* pinhole */
    float thin_prism[4];        /* pinhole */
    float max_angle;            /* fisheye + ftheta */
    int32_t ftheta_reference_poly;
    float ftheta_pixeldist_to_angle[6];
    float ftheta_angle_to_pixeldist[6];
    float ftheta_linear_cde[3];
} GrutCamera;

/* ---- 3DGUT ---------------------------------------------------------------- */
/* Mirrors the conf.render.* keys setup_3dgut.py:41-95 turns into -D macros.   */
typedef struct GutConfig {
    int32_t particle_kernel_degree;      /* 2 (default 3dgut.yaml) | 4 | 3 | 5 | 8 | 1 | 0 */
    float   particle_kernel_min_response;/* 0.0113 */
    float   particle_kernel_min_alpha;   /* 1/255  */
    float   particle_kernel_max_alpha;   /* 0.99   */
    float   min_transmittance;           /* 1e-4   */
    int32_t particle_radiance_sph_degree;/* max degree of the SH buffer: 3 -> 16 coeffs */
    int32_t enable_hitcounts;
    int32_t enable_kernel_timings;
    /* render.splat.* */
    float   ut_alpha, ut_beta, ut_kappa; /* 1, 2, 0 */
    float   ut_in_image_margin_factor;   /* 0.1 */
    int32_t ut_require_all_sigma_points_valid; /* must be 0 (threedgut.cuh:78) */
    int32_t n_rolling_shutter_iterations;/* 5 */
    int32_t k_buffer_size;               /* 0 = unsorted compositing */
    int32_t global_z_order;              /* 1 */
    int32_t rect_bounding;               /* 1 */
    int32_t tight_opacity_bounding;      /* 1 */
    int32_t tile_based_culling;          /* 1 */
    /* fp16 feature I/O (setup_3dgut.py:60-61 PARTICLE_FEATURE_HALF / FEATURE_OUTPUT_HALF; defaults 0).
     * particle_feature_half: particle_sph is [N, 3*(deg+1)^2] IEEE half (splatRaster.cpp:90-98 converts the model's tensor per call);
     *   everything is computed in fp32 from the rounded coefficients, gradients stay fp32.
     * feature_output_half: out_feat_density (and feat_density of the backward) is [H,W,4] IEEE half (rayPayload.cuh:176-186,
     *   rayPayloadBackward.cuh:50-58); GutFrame::out_features / out_opacity must then be NULL.  Hit distance / count stay fp32. */
    int32_t particle_feature_half;
    int32_t feature_output_half;
    /* Neural harmonic features (model.feature_type = nht; setup_3dgut.py:47-57, threedgrut/model/features.py:133-175; any k_buffer_size since round 6: with k_buffer_size > 0 the sorted hit buffer sits in front of the feature integration, gutKBufferRenderer.cuh:158-225).  feature_transform_type 1: `particle_sph` is the per-particle feature buffer [N, particle_feature_dim] (fp32 or
     * half), features are interpolated per hit at the canonical intersection (gutKBufferRenderer.cuh:199-225,
     * neuralHarmonicFeaturesParticle.slang:146-196) and out_feat_density is [H, W, ray_feature_dim + 1] with
     * ray_feature_dim = interp_point_feature_dim x (2 x num_frequencies for sincos | num_frequencies for siren | 1), at most 32;
     * GutFrame::out_features / out_opacity must be NULL.  Backward: gut_backward only (gutKBufferRenderer.cuh:546-641): grad_feat_density is
     * [H, W, ray_feature_dim + 1], grad_particle_sph receives the feature buffer's gradient [N, particle_feature_dim] fp32, both gradient
     * outputs are fully written; gut_backward_unpacked / _factored return GRUT_ERR_UNSUPPORTED.  All zero = SH radiance. */
    int32_t feature_transform_type;              /* 0 SH radiance (default), 1 neural harmonic features */
    int32_t particle_feature_dim;                /* K: floats per particle (nht_features.dim, 48) */
    int32_t interp_point_feature_dim;            /* K / interpolation points (12) */
    int32_t feature_interpolation_support;       /* 0 centre (1 point), 1 tetrahedra (4 points, barycentric) */
    int32_t feature_activation_type;             /* 0 none, 1 siren, 2 sincos, 3 relu */
    int32_t feature_activation_num_frequencies;  /* (1) */
} GutConfig;

/* Per-call frame description: SplatRaster::trace's non-tensor arguments. */
typedef struct GutFrame {
    uint32_t   frame_id;
    int32_t    n_active_features;  /* active SH degree (model.n_active_features) */
    uint32_t   num_particles;
    int32_t    width, height;      /* rays are [height, width, 3] */
    GrutCamera camera;
    float      pose_start[7];      /* world->sensor, [t(3), q(x,y,z,w)] (tracer.py:359-380) */
    float      pose_end[7];
    /* Optional: camera-to-world matrices (row-major 4x4 f32) in DEVICE memory.  When device_T_to_world is non-NULL the
     * library derives the sensor poses on the GPU (same math as tracer.py:359-423) and ignores pose_start / pose_end, so
     * the caller never has to read the pose back to the host (the reference's `.cpu()` there drains the stream). */
    const float* device_T_to_world;
    const float* device_T_to_world_end; /* NULL: static sensor (end = start) */
    /* Optional extra outputs of gut_forward (device pointers, may be NULL): the radiance [H,W,3] and the opacity [H,W,1] as
     * separate CONTIGUOUS tensors next to out_feat_density [H,W,4] — what the reference plugin returns as `pred_features` /
     * `pred_opacity` after `.contiguous()` (tracer.py:334-337), written by the compositing kernel itself. */
    float* out_features;
    float* out_opacity;
} GutFrame;

/* Measured work of the last forward (for the roofline byte model). */
typedef struct GutStats {
    uint32_t num_particles;      /* N  */
    uint32_t num_visible;        /* Nv: particles with >=1 tile */
    uint64_t num_intersections;  /* I  */
    uint32_t num_tiles;
    uint32_t key_bits;           /* bits of the tile part of the sort key */
    /* Work the two compositing sweeps actually did in the last frame, counted on the device when gut_profile_enable(handle, 2)
     * is in effect (0 otherwise): tile-list entries a half-tile wave EVALUATED (each one a 68-byte fetch: list entry, particle row,
     * radiance) and, of those, the entries at least one of its 128 pixels ACCEPTED (each one a 64-byte gradient slot in the
     * backward).  A sweep stops at the rays' termination, so these are far below the list length I; the roofline of the sweeps
     * is priced with them (bench.py: roofline.touched_bytes). */
    uint64_t fwd_entries_evaluated, fwd_entries_accepted;
    uint64_t bwd_entries_evaluated, bwd_entries_accepted;
} GutStats;

typedef struct GutHandle GutHandle;

int  gut_create(const GutConfig* config, GutHandle** handle);
void gut_destroy(GutHandle* handle);
int  gut_trim(GutHandle* handle);   /* release all scratch (see grut_set_allocator); the next frame allocates afresh */

/*  particle_density : [N,12] f32  {pos.xyz, density, quat.wxyz, scale.xyz, pad}
 *  particle_sph     : [N, 3*(deg+1)^2] f32, coefficient-major float3
 *  ray_origin/dir   : [H,W,3] f32, in sensor space (transformed by the inverse mid-exposure pose)
 *  out_feat_density : [H,W,4] f32  (rgb, 1-T)          fully overwritten when N > 0: rays that miss the scene box get 0
 *  out_hit_distance : [H,W,1] f32                      fully overwritten when N > 0: missing rays get 1e6 (splatRaster.cpp:213)
 *  out_hit_count    : [H,W,1] f32                      fully overwritten when N > 0 and hit counts are enabled; untouched otherwise
 *  out_visibility   : [N]     i32 (bit pattern read by the caller as float, splatRaster.cpp:215), fully overwritten
 *  With N == 0 nothing is launched and the outputs keep whatever the caller put there (the reference's zeros / 1e6). */
/*  (particle_sph / out_feat_density are void*: fp32 by default, IEEE half with GutConfig::particle_feature_half / feature_output_half) */
int gut_forward(GutHandle* handle, void* stream, const GutFrame* frame,
                const float* particle_density, const void* particle_sph,
                const float* ray_origin, const float* ray_direction,
                void* out_feat_density, float* out_hit_distance,
                float* out_hit_count, int32_t* out_visibility);

/*  grad_hit_distance     : [H,W,1] f32 or NULL when no gradient flows into the hit distance (the usual
 *                          training case: trainer.py:677-748 supervises colour/opacity only) — selects the
 *                          kernel variant without the hit-distance terms
 *  grad_particle_density : [N,12] f32 and grad_particle_sph [N, 3*(deg+1)^2] f32 are fully overwritten (no zero fill
 *                          needed): gradients are gathered per particle from per-tile-entry partials, not accumulated
 *                          with atomics, so they are bitwise reproducible from run to run */
int gut_backward(GutHandle* handle, void* stream, const GutFrame* frame,
                 const float* particle_density, const void* particle_sph,
                 const float* ray_origin, const float* ray_direction,
                 const void* feat_density, const float* grad_feat_density,
                 const float* hit_distance, const float* grad_hit_distance,
                 float* grad_particle_density, float* grad_particle_sph);

/* gut_backward with the gradient tensors in the CALLER's layout instead of the reference's packed one (new surface; the reference
 * plugin concatenates the two upstream gradients into [H,W,4] and slices the packed [N,12] result apart again,
 * threedgut_tracer/tracer.py:226-285 — two extra passes per iteration that this entry point makes unnecessary):
 *   grad_features [H,W,3], grad_opacity [H,W,1]: upstream gradients as autograd delivers them; either may be NULL (= zero);
 *   grad_positions [N,3], grad_density [N,1], grad_rotation [N,4] (16-byte aligned), grad_scale [N,3]: fully overwritten.
 * Same preconditions, errors and results (bit for bit) as gut_backward; k_buffer_size > 0 is GRUT_ERR_UNSUPPORTED here. */
typedef struct GutGradIO {
    const float* grad_features;
    const float* grad_opacity;
    float* grad_positions;
    float* grad_density;
    float* grad_rotation;
    float* grad_scale;
} GutGradIO;
int gut_backward_unpacked(GutHandle* handle, void* stream, const GutFrame* frame,
                          const float* particle_density, const void* particle_sph,
                          const float* ray_origin, const float* ray_direction,
                          const void* feat_density, const float* hit_distance, const float* grad_hit_distance,
                          const GutGradIO* io, float* grad_particle_sph);

/* ---- view-sharded data parallelism: the radiance gradient in factored form (new surface; the reference is
 * single-GPU) ------------------------------------------------------------------------------------------------------------------
 * With one view per GPU, the gradient exchange of a step is dominated by grad_particle_sph: 3*(deg+1)^2 floats per particle
 * (192 B at degree 3 of the 236 B total) that are, per view, the outer product of two small factors — the SH basis at the
 * particle's view direction (a function of the particle and the sensor position alone) and the 3-float radiance gradient
 * behind the clamp (GUTProjector::evalBackward, gutProjector.cuh:390-431).  gut_backward_factored is gut_backward without the
 * expansion: it returns that view-specific factor,
 *   grad_radiance [N+1,3] f32, fully overwritten: rows 0..N-1 = dL/d(unclamped per-particle radiance) of this view (0 where the
 *                 radiance was clamped or the particle was not rendered), row N = the view's sensor position in world space,
 * and grad_particle_density exactly as gut_backward does (including the view-direction part of the position gradient).
 * Ranks gather the factors (12 B per particle and view over the links instead of 192 B through an all-reduce) and every rank
 * rebuilds the sum over views locally with grut_sph_grad_from_views:
 *   grad_particle_sph[p][k] = scale * sum_v basis_k(normalize(position_p - sensor_v)) * grad_radiance_v[p],  k < (n_active+1)^2
 * view_factors [num_views, N+1, 3] (the gathered grad_radiance buffers, in rank order: every rank adds the views in the same
 * order, so replicas stay bitwise identical); positions: N rows of position_stride floats whose first three are the particle
 * position (3 for a [N,3] tensor, 12 for particle_density rows); scale: 1 for a sum over views, 1/num_views for a mean.
 * With num_views = 1 and scale = 1 the result is bit for bit what gut_backward writes.  Same preconditions and errors as
 * gut_backward (forward context on the same stream); with N == 0 nothing is launched and nothing is written. */
int gut_backward_factored(GutHandle* handle, void* stream, const GutFrame* frame,
                          const float* particle_density, const void* particle_sph,
                          const float* ray_origin, const float* ray_direction,
                          const void* feat_density, const float* grad_feat_density,
                          const float* hit_distance, const float* grad_hit_distance,
                          float* grad_particle_density, float* grad_radiance);

/* gut_backward_factored with the gradient finalisation cut into `num_chunks` particle ranges (starting at multiples of 128): after the
 * launches of each chunk the library calls on_chunk(user, chunk, first_particle, num_particles) ON THE CALLING THREAD; rows [first, first +
 * num) of grad_particle_density and grad_radiance are then complete in stream order, and the caller may enqueue that chunk's collectives
 * (3dgrut_amd/dp.py: all-reduce of the packed rows, all-gather of the view factors), which run under the next chunk's kernels.  Row N of
 * grad_radiance (the sensor position) is written with the first chunk.  Results are bit for bit gut_backward_factored's. */
typedef void (*GrutChunkFn)(void* user, uint32_t chunk, uint32_t first_particle, uint32_t num_particles);
int gut_backward_factored_chunked(GutHandle* handle, void* stream, const GutFrame* frame,
                                  const float* particle_density, const void* particle_sph,
                                  const float* ray_origin, const float* ray_direction,
                                  const void* feat_density, const float* grad_feat_density,
                                  const float* hit_distance, const float* grad_hit_distance,
                                  float* grad_particle_density, float* grad_radiance,
                                  uint32_t num_chunks, GrutChunkFn on_chunk, void* user);
int grut_sph_grad_from_views(void* stream, uint32_t num_particles, uint32_t num_views, const float* view_factors,
                             const float* positions, uint32_t position_stride, int32_t n_active_features, int32_t sph_degree,
                             float scale, float* grad_particle_sph);

/* average ms of the forward / backward launches since the last call (-1 if none). Synchronises. */
int gut_timings(GutHandle* handle, float* forward_ms, float* backward_ms);
int gut_stats(GutHandle* handle, GutStats* stats);

/* Per-stage device time (hipEvents on the launch stream), for the roofline report.  While enabled, every
 * gut_forward / gut_backward brackets each stage with an event pair; gut_profile_read synchronises, returns the
 * average ms per stage since the last read (-1 where a stage did not run) and resets the accumulators. */
enum { GUT_STAGE_PROJECT = 0, GUT_STAGE_DEPTH_SORT, GUT_STAGE_SCAN, GUT_STAGE_EXPAND, GUT_STAGE_TILE_SORT,
       GUT_STAGE_TILE_RANGES, GUT_STAGE_RENDER_FWD, GUT_STAGE_RENDER_BWD, GUT_STAGE_PROJECT_BWD, GUT_NUM_STAGES };
int gut_profile_enable(GutHandle* handle, int enable);
/* Restrict the event pairs to the stages whose bit (1 << GUT_STAGE_*) is set (default: all).  Two event records between adjacent
 * stages leave ~10 us of idle stream on MI355X; timing only the kernel of interest keeps the rest of the frame back to back. */
int gut_profile_select(GutHandle* handle, uint32_t stage_mask);
int gut_profile_read(GutHandle* handle, float* stage_ms /* [GUT_NUM_STAGES] */);

/* ---- stage-level entry points (parity tests drive each stage alone) ------ */
/* Stable LSD radix sort of (key,value) pairs on key bits [begin_bit,end_bit). tmp buffers sized n. */
int grut_sort_pairs_u32(void* stream, uint32_t n, int begin_bit, int end_bit,
                        uint32_t* keys, uint32_t* values,
                        uint32_t* keys_tmp, uint32_t* values_tmp,
                        void* scratch, uint64_t scratch_bytes,
                        uint32_t** sorted_keys, uint32_t** sorted_values);
uint64_t grut_sort_scratch_bytes(uint32_t n);
/* inclusive prefix sum of n u32 (cub::DeviceScan::InclusiveSum, gutRenderer.cu:302-310) */
int grut_inclusive_scan_u32(void* stream, uint32_t n, const uint32_t* in, uint32_t* out,
                            void* scratch, uint64_t scratch_bytes);
uint64_t grut_scan_scratch_bytes(uint32_t n);

/* Copies the binning products of the last gut_forward to caller DEVICE buffers (any may be NULL):
 * tiles_count[N] u32, proj_pos[N,2], conic_opacity[N,4], extent[N,2], depth[N], rgb[N,3],
 * sorted_particle_idx[I] u32, tile_ranges[tiles,2] u32. */
/* Diagnostics of an instrumented frame (gut_profile_enable level 2): the raw counter block to a caller DEVICE buffer — words 0..3 =
 * GutStats' four entry counters, words 16 + 4 b .. 19 + 4 b = lifetime and start (100 MHz ticks), accepted << 32 | evaluated entries and
 * tile-list length of forward-sweep workgroup b. */
int gut_debug_fetch_work(GutHandle* handle, void* stream, unsigned long long* out, uint64_t count);
/* HOST function, no device work: the world->sensor pose [t(3), q(x,y,z,w)] the library derives from a camera-to-world matrix handed
 * over as GutFrame::device_T_to_world (row-major 4x4, rows 0-2 read) - the host twin of the device code, same arithmetic: float64
 * general inverse, one rounding to float32, float32 quaternion (threedgut_tracer/tracer.py:88-136, 359-380, 413-423). */
int grut_debug_pose_from_c2w(const float* c2w16, float* out7);
/* The whole per-frame pose block the kernels read (47 floats: start R[9] t[3] q[4], end t[3] q[4], mid-exposure world->sensor
 * R[9] t[3], sensor->world R[9] t[3]; sensors.h:44-73, gutRenderer.cu:266-267) from camera-to-world matrices: on_device != 0 runs the
 * device code on DEVICE matrices and synchronises the stream, on_device == 0 its host twin on HOST matrices.  T_end may be NULL. */
int grut_debug_frame_poses(void* stream, int on_device, const float* T_start, const float* T_end, float* host_out47);
int gut_debug_fetch(GutHandle* handle, void* stream,
                    uint32_t* tiles_count, float* proj_pos, float* conic_opacity, float* extent,
                    float* depth, float* rgb, uint32_t* sorted_particle_idx, uint32_t* tile_ranges);

/* ---- 3DGRT ----------------------------------------------------------------- */
typedef struct GrtConfig {
    int32_t particle_kernel_degree;        /* 4 (3dgrt.yaml) */
    float   particle_kernel_min_response;  /* 0.0113 */
    float   particle_kernel_min_alpha;     /* 1/255 */
    float   particle_kernel_max_alpha;     /* 0.99 */
    int32_t particle_kernel_density_clamping; /* 1 */
    int32_t particle_radiance_sph_degree;  /* 3 */
    int32_t enable_normals;
    int32_t enable_hitcounts;
    int32_t enable_kernel_timings;
    int32_t max_hits_per_trace;            /* 16 (pipelineParameters.h:83) */
    /* fp16 feature I/O (setup_3dgrt.py:42-43; 3dgrt/pipelineParameters.h:24-46): particle_sph as IEEE half / out_features (and the
     * `features` input of the backward) as [H,W,3] IEEE half; arithmetic and gradients stay fp32 */
    int32_t particle_feature_half;
    int32_t feature_output_half;
    /* Neural harmonic features (model.feature_type = nht with render.pipeline_type referenceSlang / referenceSlangBwd:
     * referenceSlangOptix.cu:103-186, referenceSlangBwdOptix.cu:70-185), same fields and meaning as in GutConfig: particle_sph is the
     * feature buffer [N, particle_feature_dim], out_features / features are [H, W, ray_feature_dim] (at most 32), grad_particle_sph the
     * feature buffer's gradient; enable_normals must be 0.  All zero = SH radiance (the `reference` pipeline). */
    int32_t feature_transform_type;
    int32_t particle_feature_dim;
    int32_t interp_point_feature_dim;
    int32_t feature_interpolation_support;
    int32_t feature_activation_type;
    int32_t feature_activation_num_frequencies;
    /* render.primitive_type (optixTracer.cpp:176-201): the proxy geometry a particle is traced through.  GRUT_PRIM_INSTANCES: the unit cube
     * under the particle's instance transform, hit distance = the point of maximum response (intersectInstanceParticle).  The closed convex
     * triangle meshes of particlePrimitives.cu:63-496: the ray is offered the particle at the distance at which it ENTERS the proxy (OptiX
     * triangles with back faces culled, referenceOptix.cu:62), rays that start inside are not offered it.  GRUT_PRIM_CUSTOM: custom primitives over
     * the particles' WORLD boxes (computeGaussianEnclosingAABBKernel) with the world-space intersection program intersectCustomParticle
     * (gaussianParticles.cuh:407-441): the instances' hit point, offered to the rays that cross the world box, accepted within 3 sigma.
     * The open meshes GRUT_PRIM_TRISURFEL / GRUT_PRIM_TRIHEXA and the enclosing spheres GRUT_PRIM_SPHERE: below. */
    int32_t primitive_type;
    /* render.pipeline_type (optixTracer.cpp:246-318: the forward program's file).  GRUT_PIPELINE_REFERENCE: referenceOptix.cu (and the Slang
     * pipelines, served by the same kernels).  GRUT_PIPELINE_BARYCENTRIC_SURFELS: barycentricSurfelsOptix.cu - FORWARD ONLY (the reference ships
     * no backward program for it): trisurfel proxies (primitive_type must be GRUT_PRIM_TRISURFEL), ten hits per trace, the response evaluated
     * from the squared distance of the ray's crossing of the surfel's plane in the proxy frame (the hit triangle's barycentrics) with the
     * kernel-scaled minimum response, depth from the triangle hit distances, normals from the surfel's plane.  grt_backward returns
     * GRUT_ERR_UNSUPPORTED. */
    int32_t pipeline_type;
} GrtConfig;
enum { GRUT_PIPELINE_REFERENCE = 0, GRUT_PIPELINE_BARYCENTRIC_SURFELS = 1 };
enum { GRUT_PRIM_INSTANCES = 0, GRUT_PRIM_ICOSAHEDRON = 1, GRUT_PRIM_OCTAHEDRON = 2, GRUT_PRIM_TETRAHEDRON = 3, GRUT_PRIM_DIAMOND = 4, GRUT_PRIM_CUSTOM = 5,
       /* trisurfel (particlePrimitives.cu:155-205): two triangles per particle = the rhombus |x| + |y| <= sqrt 2 of the proxy's z = 0 plane, traced
        * WITHOUT face culling (referenceOptix.cu:62); the hit is the ray's crossing of that plane and the per-hit math takes its
        * SurfelPrimitive branches (gaussianParticles.cuh:371-400, 512-521, 558-565, 628-659). */
       GRUT_PRIM_TRISURFEL = 6,
       /* trihexa (particlePrimitives.cu:107-153): three rhombi in the proxy's coordinate planes, six triangles, back faces culled - the windings
        * make the x = 0 / y = 0 rhombi face +x / +y and the two halves of the z = 0 rhombus face opposite ways, so a ray is offered the SAME
        * particle up to three times at three distances and the programs process every offer as a hit of the particle.  Here every rhombus is a
        * proxy of its own (3 N leaves, proxy 3 i + j = plane j of particle i). */
       GRUT_PRIM_TRIHEXA = 7,
       /* sphere (optixTracer.cpp:189-190, 765-781, 823-833; computeGaussianEnclosingSphereKernel, particlePrimitives.cu:386-403): one OptiX
        * built-in sphere per particle, centre mu, radius max(scale) * kernelScale.  The built-in intersector offers the any-hit program the
        * ray's ENTRY into the sphere and - the program ignores every offer but the one that fills its payload - its EXIT as well: the particle is
        * a candidate at both roots and is processed twice when both fall into the ray's rounds (the volumetric processHit either time).  Here
        * every root is a proxy of its own (2 N leaves, proxy 2 i = entry, 2 i + 1 = exit of particle i).  NVIDIA does not publish the
        * intersector's arithmetic: the roots are those of |po + t pd|^2 = 1 in the frame scaled by the radius, operation by operation as
        * oracle/grt_oracle.c and the emulated OptiX of oracle/ref/ref_grt_emul.inl evaluate them. */
       GRUT_PRIM_SPHERE = 8 };

typedef struct GrtFrame {
    uint32_t frame_id;
    int32_t  sph_degree;          /* active SH degree */
    float    min_transmittance;
    uint32_t num_particles;
    int32_t  width, height;
    float    ray_to_world[12];    /* row-major 3x4 */
    int32_t  keep_hits_for_backward; /* forward: record the processed hits so that grt_backward of the same frame
                                      * replays them instead of traversing again (identical results) */
    const float* device_ray_to_world; /* optional: row-major 4x4 (or 3x4) matrix in DEVICE memory, read by the kernels instead of
                                       * ray_to_world — a pose that lives on the GPU needs no host copy (the reference does
                                       * rayToWorld.cpu() per call, optixTracer.cpp:931) */
} GrtFrame;

typedef struct GrtStats {
    uint32_t num_particles;
    uint32_t num_nodes;
    uint64_t nodes_visited;     /* only in instrumented launches */
    uint64_t candidates;
    uint64_t processed_hits;
    float    scene_aabb[6];
    uint64_t list_entries;      /* last forward: entries of the packet lists (0: the tree walk served the frame — rays with different origins) */
    uint64_t packet_tests;      /* only in instrumented launches: candidate tests of whole packets (list entries / leaves tested by a wave) */
    uint64_t list_batches;      /* only in instrumented launches: 64-entry batches of packet lists fetched by the trace rounds (one 64-byte record per entry) */
    uint32_t bwd_rederived_rays;   /* last replayed backward: rays whose rounds were re-derived instead of replayed (a round met more ghosts than a log chunk holds) */
    uint32_t bwd_premise_rays;     /* last replayed backward: rays for which the ghost filter's premise failed (DESIGN.md "3DGRT backward"; expected 0) */
    uint64_t bwd_atomic_instructions; /* last replayed backward: atomic-add instructions issued (one per differentiated hit or same-slot group of hits) */
    uint64_t bwd_atomic_words;     /* ... and the non-zero float words they carried (to 16): what scripts/atomic_calib.hip's rate prices */
} GrtStats;

typedef struct GrtHandle GrtHandle;

/* ---- hybrid mesh + Gaussian path tracing (BASELINE config 5) -------------------------------------------------------------------
 * Replaces HybridOptixTracer::buildMeshBVH / traceHybrid (threedgrut_playground/include/playground/hybridTracer.h:121-141) and the
 * OptiX programs of threedgrut_playground/src/kernels/cuda/playgroundKernel.cu:39-352 with materials.cuh, trace.cuh, rng.cuh under
 * them: per ray a path loop — closest triangle, the face's primitive type (none / mirror / glass / diffuse / PBR: Cook-Torrance
 * sampling with GGX importance sampling, glTF alpha modes, diffuse / emissive / metallic-roughness / normal textures, vertex tangents),
 * then the Gaussians between the ray origin and the surface with the forward program's k = 16 rounds (3dgrtTracer.cuh:137-204),
 * transmittance and path throughput carried along the whole path; the environment map is looked up along the last ray.  Forward
 * only, like the reference.  All array pointers are DEVICE pointers owned by the caller; the GrtMaterial table itself is a HOST
 * array (the library uploads it with the launch: a few hundred bytes, as HybridOptixTracer::syncMaterials does).
 * Textures: row-major [height, width, channels] f32, sampled with normalised coordinates, clamp-to-edge, bilinear — the modes
 * playground/cutexture.h:54-60 sets (CUDA filters with 8 fractional weight bits; here the float weights themselves). */
typedef struct GrtTexture {
    const float* data;                /* NULL: no texture */
    int32_t height, width, channels;
} GrtTexture;
typedef struct GrtMaterial {          /* PBRMaterial, playground/pipelineParameters.h:26-52 */
    GrtTexture diffuse, emissive, metallic_roughness, normal;   /* 4, 4, 2, 4 channels */
    float diffuse_factor[4], emissive_factor[3];
    float metallic_factor, roughness_factor, transmission_factor, ior, alpha_cutoff;
    uint32_t alpha_mode;              /* GltfAlphaMode (pipelineDefinitions.h:42-48): 0 opaque, 1 blend, 2 mask */
} GrtMaterial;
typedef struct GrtMesh {
    uint32_t num_vertices, num_faces;
    const float*   vertices;            /* [V,3] */
    const int32_t* triangles;           /* [F,3] vertex indices */
    const float*   vertex_normals;      /* [V,3] (needed with playground_opts bit 0) */
    const float*   vertex_tangents;     /* [V,3] or NULL */
    const uint8_t* vertex_has_tangents; /* [V]   or NULL (no precomputed tangents) */
    const int32_t* prim_type;           /* [F] PlaygroundPrimitiveTypes (pipelineDefinitions.h:18-24): 0 none, 1 mirror, 2 glass, 3 diffuse, 4 PBR */
    const float*   mat_uv;              /* [F,3,2] texture coordinates per face corner, or NULL (0) */
    const int32_t* mat_id;              /* [F] row of `materials`, or NULL (0) */
    const float*   refractive_index;    /* [F] */
    uint32_t num_materials;
    const GrtMaterial* materials;       /* HOST array [num_materials] (needed when a diffuse or PBR face exists) */
    GrtTexture envmap;                  /* [EH,EW,4]; data == NULL: black */
    float envmap_offset[2];             /* rotates the environment (trace.cuh:233-257) */
} GrtMesh;
typedef struct GrtHybridOptions {
    uint32_t playground_opts;         /* PlaygroundRenderOptions: bit 0 smooth normals, bit 1 disable Gaussian tracing, bit 2 disable PBR textures */
    uint32_t max_pbr_bounces;         /* the path loop runs while pbrNumBounces < max_pbr_bounces */
    uint32_t frame_number;            /* seeds the per-pixel random streams (rng.cuh; playgroundKernel.cu:59, materials.cuh:224-228) */
} GrtHybridOptions;
/* rebuild = 0 with allow_update != 0 and an unchanged face count refits the boxes of the existing tree (OPTIX_BUILD_OPERATION_UPDATE,
 * hybridTracer.cpp buildMeshBVH); anything else builds from scratch */
int grt_build_mesh_bvh(GrtHandle* handle, void* stream, uint32_t num_vertices, const float* vertices, uint32_t num_faces,
                       const int32_t* triangles, int rebuild, int allow_update);
/* rays [H,W,3] in ray space (frame->ray_to_world applies); ray_max_t [H,W] or NULL; out_radiance [H,W,3], out_opacity [H,W,1] fully
 * written; out_last_ray [H,W,6] (world-space origin + direction of the last segment: what the reference writes back into its ray
 * buffers, trace.cuh:158-173) and out_bounces [H,W] (mirror bounces) may be NULL. */
/* (particle_sph: IEEE half with GrtConfig::particle_feature_half; the four outputs are always fp32) */
int grt_trace_hybrid(GrtHandle* handle, void* stream, const GrtFrame* frame, const float* particle_density, const void* particle_sph,
                     const float* ray_origin, const float* ray_direction, const float* ray_max_t, const GrtMesh* mesh,
                     const GrtHybridOptions* options, float* out_radiance, float* out_opacity, float* out_last_ray, uint32_t* out_bounces);

int  grt_create(const GrtConfig* config, GrtHandle** handle);
int  grt_trim(GrtHandle* handle);
void grt_destroy(GrtHandle* handle);

/* positions[N,3], rotations[N,4] wxyz normalised, scales[N,3], densities[N] — activated values */
int grt_build_bvh(GrtHandle* handle, void* stream, uint32_t num_particles,
                  const float* positions, const float* rotations,
                  const float* scales, const float* densities,
                  int rebuild, int allow_update);

/*  out_features [H,W,3], out_density [H,W,1], out_hit_distance [H,W,2] (integrated depth, last hit t),
 *  out_normals [H,W,3], out_hits_count [H,W,1], out_visibility [N] i32 — all must arrive zero-filled.
 *  Replaces OptixTracer::trace (optixTracer.h:150-163, optixTracer.cpp:890-1000).  When every ray of the frame starts at the same
 *  point (decided on the device from ray_origin) the candidates of each 8x8 ray packet are binned once per frame and the k = 16 trace
 *  rounds scan sorted per-packet lists instead of walking the BVH (identical results; GrtStats::list_entries tells which path ran; the
 *  call then waits once for the list size — a 4-byte read-back — before it enqueues the trace).  Environment GRUT_GRT_NO_LISTS=1
 *  forces the tree walk. */
/*  (particle_sph / out_features are void*: fp32 by default, IEEE half with GrtConfig::particle_feature_half / feature_output_half) */
int grt_forward(GrtHandle* handle, void* stream, const GrtFrame* frame,
                const float* particle_density, const void* particle_sph,
                const float* ray_origin, const float* ray_direction,
                void* out_features, float* out_density, float* out_hit_distance,
                float* out_normals, float* out_hits_count, int32_t* out_visibility);

int grt_backward(GrtHandle* handle, void* stream, const GrtFrame* frame,
                 const float* particle_density, const void* particle_sph,
                 const float* ray_origin, const float* ray_direction,
                 const void* features, const float* density, const float* hit_distance, const float* normals,
                 const float* grad_features, const float* grad_density,
                 const float* grad_hit_distance, const float* grad_normals,
                 float* grad_particle_density, float* grad_particle_sph);

/* grt_forward that also records, per ray, the particles it processed in order (the BVH hit-order parity test):
 * hit_ids [H*W, capacity] u32, hit_counts [H*W] u32 (counts may exceed capacity; only the first `capacity` are stored) */
int grt_debug_forward_hits(GrtHandle* handle, void* stream, const GrtFrame* frame,
                           const float* particle_density, const void* particle_sph,
                           const float* ray_origin, const float* ray_direction,
                           void* out_features, float* out_density, float* out_hit_distance,
                           float* out_normals, float* out_hits_count, int32_t* out_visibility,
                           uint32_t* hit_ids, uint32_t* hit_counts, uint32_t capacity);
/* Parity aid: until called again with NULLs, every grt_backward also writes, per ray (caller DEVICE buffers of W*H entries, zero-filled
 * by the caller), how many hits it differentiated and an order-independent signature of which particles they were
 * (sum of particle * 0x9E3779B97F4A7C15 + 1, mod 2^64) — the tests compare the replayed backward with the re-derived one ray by ray. */
int grt_debug_backward_signature(GrtHandle* handle, unsigned long long* ray_signature, uint32_t* ray_hit_count);
/* proxy instance records of the last build: [N,12] f32 = rows of W = diag(1/kscl) R^T, then mu (object ray: o' = W (o - mu)) */
int grt_debug_fetch_instances(GrtHandle* handle, void* stream, float* instances);
/* GRUT_PRIM_CUSTOM: [N,8] f32 = the particle's world box (min xyz, max xyz; computeGaussianEnclosingAABBKernel, particlePrimitives.cu:498-541),
 * kernelScale^2, 0 - what the candidate test of the custom primitives reads.  GRUT_ERR_NOT_READY for another primitive type. */
int grt_debug_fetch_custom_boxes(GrtHandle* handle, void* stream, float* box8);
/* The per-packet candidate lists of the last train-mode forward (GrtStats::list_entries of them) to caller DEVICE buffers: ranges [blocks,2]
 * ([first, last) of each 8x8 ray packet, row-major), entries [list_entries] particle ids (top bit: internal flag).  GRUT_ERR_NOT_READY when
 * the tree walk served the frame or the capacity is too small.  The parity tests hand them to the CPU checker as a candidate prefilter. */
int grt_debug_fetch_lists(GrtHandle* handle, void* stream, uint32_t* ranges, uint32_t* entries, uint64_t entry_capacity);

int grt_timings(GrtHandle* handle, float* forward_ms, float* backward_ms, float* build_ms);
int grt_stats(GrtHandle* handle, GrtStats* stats);
/* Diagnostics of an instrumented forward (environment GRUT_GRT_COUNT=1): the raw counter block to a caller DEVICE buffer — words 0..11
 * the totals grt_stats reports, then per launched workgroup {start, lifetime (10 ns ticks of the chip-wide counter), node visits << 32 |
 * leaf visits}.  Development aid (scripts/diag_grt_balance.py). */
int grt_debug_fetch_work(GrtHandle* handle, void* stream, unsigned long long* out, uint64_t count);

/* ---- parameter marshalling -------------------------------------------------- */
/* [N,3] positions, [N,1] density, [N,4] rotation (wxyz), [N,3] scale (all contiguous fp32, DEVICE) -> [N,12] ParticleDensity
 * rows {position, density, quaternion, scale, 0}: the torch.cat of threedgut_tracer/tracer.py:178 and
 * threedgrt_tracer/tracer.py:93-96, in one pass. */
int grut_pack_particles(void* stream, uint32_t num_particles, const float* positions, const float* density,
                        const float* rotation, const float* scale, float* particle_density);

/* The inverse for gradients: packed [N,12] -> [N,3] positions, [N,1] density, [N,4] rotation, [N,3] scale, contiguous
 * (what tracer.py:268-285 hands to autograd as strided slices). */
int grut_unpack_particle_grads(void* stream, uint32_t num_particles, const float* grad_particle_density, float* grad_positions,
                               float* grad_density, float* grad_rotation, float* grad_scale);

/* The same packing fused with the model's activations (threedgrut/model/model.py:102-118 with the defaults of
 * configs/base_gs.yaml:77-78 and model.py:241): density = sigmoid(raw), scale = exp(raw), rotation = normalize(raw).
 * Replaces three elementwise passes + the torch.cat per call. */
int grut_activate_pack(void* stream, uint32_t num_particles, const float* positions, const float* raw_density,
                       const float* raw_rotation, const float* raw_scale, float* particle_density);
/* Chain rule of the above: the renderer's packed gradient [N,12] -> gradients of the four RAW tensors (all fully written). */
int grut_activate_pack_backward(void* stream, uint32_t num_particles, const float* raw_density, const float* raw_rotation,
                                const float* raw_scale, const float* grad_particle_density, float* grad_positions,
                                float* grad_raw_density, float* grad_raw_rotation, float* grad_raw_scale);

/* ---- optimizer step --------------------------------------- */
/* One parameter group of SelectiveAdam (threedgrut/optimizers/__init__.py:85-124): contiguous fp32 [num_rows, row_width]
 * DEVICE tensors, 16-byte aligned. */
typedef struct GrutAdamGroup {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    uint32_t row_width;
    float lr, beta1, beta2, eps;
} GrutAdamGroup;
enum { GRUT_VIS_NONE = 0,        /* every row is updated */
       GRUT_VIS_BOOL_U8 = 1,     /* visibility = bool[num_rows] (what `visibility.bool().squeeze()` yields) */
       GRUT_VIS_INT32 = 2,       /* visibility = int32[num_rows], visible iff != 0 (gut_forward's out_visibility) */
       GRUT_VIS_FLOAT_BITS = 3   /* visibility = float[num_rows], visible iff != 0.0f (the tracers' `mog_visibility`) */ };
/* Replaces selective_adam_update_launch (threedgrut/optimizers/optimizers.cu:79-109, kernel :49-77), for ALL groups in
 * one launch: rows whose flag is zero keep parameter and moments; visible rows get
 * m = b1 m + (1-b1) g; v = b2 v + (1-b2) g g; param -= lr m / (sqrt(v) + eps)   (no bias correction, as the reference). */
int grut_selective_adam_update(void* stream, const GrutAdamGroup* groups, int num_groups, uint32_t num_rows,
                               const void* visibility, int visibility_kind);

/* human-readable text of the last error raised on the calling thread */
const char* grut_last_error(void);
/* ABI version; bumped whenever a struct above changes */
int grut_abi_version(void);

/* Scratch memory.  Every per-frame buffer of the two renderers is grow-only device scratch (the role of the reference's CudaBuffer,
 * src/cudaBuffer.cpp:44-60, which calls cudaMalloc / cudaFree).  By default it comes from hipMalloc / hipFree; a host that has its own
 * device allocator — PyTorch's caching allocator in the plugins — installs it here BEFORE creating handles: alloc_fn returns device
 * memory usable on the stream of the API call it is made from (NULL = failure), free_fn takes it back.  Process-wide; NULL, NULL
 * restores hipMalloc.  gut_trim / grt_trim release everything a handle holds (the handle stays valid; 3DGRT needs its BVH rebuilt). */
typedef void* (*GrutAllocFn)(void* user, uint64_t bytes);
typedef void  (*GrutFreeFn)(void* user, void* ptr);
int grut_set_allocator(GrutAllocFn alloc_fn, GrutFreeFn free_fn, void* user);

#ifdef __cplusplus
}
#endif
#endif /* GRUT_AMD_H */

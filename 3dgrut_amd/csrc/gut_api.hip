// gut_api.hip — host orchestration of the 3DGUT path behind the C-ABI (include/grut_amd.h).
// Plays the role of SplatRaster + GUTRenderer (threedgut_tracer/src/splatRaster.cpp:184-350,
// src/gutRenderer.cu:241-520) without libtorch: all I/O buffers belong to the caller.
#include <vector>

#include "gut_internal.hpp"

namespace grut {

static thread_local char g_last_error[512] = "";
ScratchAllocator& scratch_allocator() {
    static ScratchAllocator a;
    return a;
}
void set_last_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

static uint32_t bits_for(uint32_t n) {  // smallest b with (1<<b) > n  -> the all-ones pad key never aliases a tile
    uint32_t b = 1;
    while ((1ull << b) <= n) ++b;
    return b;
}

}  // namespace grut

using namespace grut;

struct GutHandle {
    GutConfig cfg;
    int device = -1;
    // per-particle scratch
    DeviceBuffer tiles_count, proj_pos, conic_opacity, extent, depth, rgb, depth_key, particle_idx;
    DeviceBuffer depth_key_tmp, particle_idx_tmp, offsets, sort_scratch, scan_scratch, counters;
    DeviceBuffer rec64;   // [N][4] float4: the per-particle 64-byte records of the direct tile lists (GutParams::rec64)
    DeviceBuffer part_offset, pos_particle, grad_partial, grad_flag, g_rgb, poses_dev, walk8;
    // per-intersection scratch
    DeviceBuffer tile_keys, tile_vals, tile_keys_tmp, tile_vals_tmp, tile_sort_scratch, ranges;
    DeviceBuffer ck_tc, ck_d, ck_reached, ck_boundary_tile, ck_nht;
    GutCheckpoints checkpoints;
    uint32_t* host_counters = nullptr;  // pinned: [0] = I (last offset), [1] = Nv
    hipEvent_t count_event = nullptr;
    // forward context consumed by backward (role of GutRenderForwardContext)
    bool have_forward = false;
    hipStream_t fwd_stream = nullptr;
    // identity of the forward the context belongs to: the backward must be handed the same frame and the very buffers that
    // forward saw (the packed particle buffer is a fresh allocation per forward that the caller keeps alive until the backward,
    // so its address is a per-forward token).  The reference keeps ONE context too (gutRenderer.cu:436-440) and silently
    // differentiates the wrong frame when two forwards precede their backwards; here that is an error.
    const void* fwd_density = nullptr;
    const void* fwd_ray_o = nullptr;
    const void* fwd_ray_d = nullptr;
    uint32_t fwd_frame_id = 0;
    GutParams params;
    uint32_t num_intersections = 0;
    uint32_t tile_capacity = 0;  // entries the per-intersection scratch can hold (0 until the first frame sized it)
    uint32_t ck_boundaries_capacity = 0;
    uint32_t* sorted_pos = nullptr;  // sorted expansion positions: points into tile_vals or tile_vals_tmp
    uint32_t* sorted_tile_keys = nullptr;
    uint32_t* rank_to_particle = nullptr;
    GutStats stats;
    EventTimer fwd_timer, bwd_timer;
    // per-stage profiling (gut_profile_enable)
    bool profile = false;
    bool count_work = false;            // gut_profile_enable(handle, 2): the sweeps count their evaluated / accepted entries
    DeviceBuffer work_counters;
    unsigned long long work_host[4] = {0, 0, 0, 0};
    bool work_pending = false;
    static constexpr int kProfRing = 64;
    hipEvent_t prof_ev[kProfRing][GUT_NUM_STAGES][2];
    bool prof_used[kProfRing][GUT_NUM_STAGES];
    int prof_slot = 0, prof_created = 0;
    int prof_fwd_slot = 0;

    uint32_t profile_mask = 0xFFFFFFFFu;   // gut_profile_select: stages that get an event pair while profiling is on
    int stage_begin(int stage, hipStream_t s, int slot) {
        if (!profile || !((profile_mask >> stage) & 1u)) return GRUT_OK;
        GRUT_HIP(hipEventRecord(prof_ev[slot][stage][0], s));
        return GRUT_OK;
    }
    int stage_end(int stage, hipStream_t s, int slot) {
        if (!profile || !((profile_mask >> stage) & 1u)) return GRUT_OK;
        GRUT_HIP(hipEventRecord(prof_ev[slot][stage][1], s));
        prof_used[slot][stage] = true;
        return GRUT_OK;
    }
};

static int validate_config(const GutConfig& c) {
    GRUT_REQUIRE(c.ut_require_all_sigma_points_valid == 0, "ut_require_all_sigma_points_valid must be false (threedgut.cuh:78)");
    GRUT_REQUIRE(c.particle_radiance_sph_degree >= 0 && c.particle_radiance_sph_degree <= 3, "sph degree must be in [0,3]");
    if (c.k_buffer_size != 0 && c.k_buffer_size != 4 && c.k_buffer_size != 8 && c.k_buffer_size != 16) {
        set_last_error("k_buffer_size=%d: the hit buffer is instantiated for 0 (unsorted), 4, 8 and 16 entries", c.k_buffer_size);
        return GRUT_ERR_UNSUPPORTED;
    }
    if (c.feature_transform_type != 0) {
        GRUT_REQUIRE(c.feature_transform_type == 1, "feature_transform_type %d: 0 (SH) or 1 (neural harmonic features)", c.feature_transform_type);
        // (k_buffer_size > 0 with features: the sorted hit buffer in front of the feature integration, gutKBufferRenderer.cuh:158-225 - round 6)
        GRUT_REQUIRE(c.feature_interpolation_support == 0 || c.feature_interpolation_support == 1, "feature_interpolation_support must be 0 (centre) or 1 (tetrahedra)");
        GRUT_REQUIRE(c.feature_activation_type >= 0 && c.feature_activation_type <= 3, "feature_activation_type must be 0..3");
        const int points = c.feature_interpolation_support == 1 ? 4 : 1;
        GRUT_REQUIRE(c.interp_point_feature_dim >= 1 && c.interp_point_feature_dim <= 16 && c.particle_feature_dim == points * c.interp_point_feature_dim,
                     "particle_feature_dim %d must be %d x interp_point_feature_dim (1..16)", c.particle_feature_dim, points);
        const int nf = c.feature_activation_num_frequencies;
        const int nr = c.interp_point_feature_dim * (c.feature_activation_type == 2 ? 2 * nf : (c.feature_activation_type == 1 ? nf : 1));
        GRUT_REQUIRE(nf >= 1 && nr >= 1 && nr <= 32, "ray feature dim %d: 1..32 supported", nr);
    }
    const int d = c.particle_kernel_degree;
    GRUT_REQUIRE(d == 0 || d == 1 || d == 2 || d == 3 || d == 4 || d == 5 || d == 8, "unsupported particle_kernel_degree %d", d);
    return GRUT_OK;
}

static GutParams make_params(const GutConfig& c, const GutFrame& f) {
    GutParams P;
    memset(&P, 0, sizeof(P));
    P.degree = c.particle_kernel_degree;
    P.min_response = c.particle_kernel_min_response;
    P.min_alpha = c.particle_kernel_min_alpha;
    P.max_alpha = c.particle_kernel_max_alpha;
    P.min_transmittance = c.min_transmittance;
    P.n_active = f.n_active_features < c.particle_radiance_sph_degree ? f.n_active_features : c.particle_radiance_sph_degree;
    if (P.n_active < 0) P.n_active = 0;
    P.ncoef = (c.particle_radiance_sph_degree + 1) * (c.particle_radiance_sph_degree + 1);
    P.hitcounts = c.enable_hitcounts;
    const float D = 3.f;
    const float lambda = c.ut_alpha * c.ut_alpha * (D + c.ut_kappa) - D;  // gutProjector.cuh:150
    P.ut_w0m = lambda / (D + lambda);
    P.ut_wi = 1.f / (2.f * (D + lambda));
    P.ut_w0c = lambda / (D + lambda) + (1.f - c.ut_alpha * c.ut_alpha + c.ut_beta);
    P.ut_delta = sqrtf(c.ut_alpha * c.ut_alpha * (D + c.ut_kappa));  // setup_3dgut.py:44
    P.ut_margin = c.ut_in_image_margin_factor;
    P.ut_require_all = c.ut_require_all_sigma_points_valid;
    P.n_rs_iter = c.n_rolling_shutter_iterations;
    P.k_buffer = c.k_buffer_size;
    P.global_z = c.global_z_order;
    P.rect_bounding = c.rect_bounding;
    P.tight_opacity = c.tight_opacity_bounding;
    P.tile_culling = c.tile_based_culling;
    P.W = f.width;
    P.H = f.height;
    P.gx = (f.width + 15) / 16;
    P.gy = (f.height + 15) / 16;
    P.N = f.num_particles;
    P.cam = f.camera;
    P.poses = make_frame_poses(f.pose_start, f.pose_end);
    P.poses_dev = nullptr;
    P.work = nullptr;
    P.out_features = f.out_features;
    P.out_opacity = f.out_opacity;
    P.sph_half = c.particle_feature_half;
    P.out_half = c.feature_output_half;
    P.nht = c.feature_transform_type;
    if (P.nht) {
        P.nht_k = c.particle_feature_dim; P.nht_ipd = c.interp_point_feature_dim; P.nht_support = c.feature_interpolation_support;
        P.nht_act = c.feature_activation_type; P.nht_nf = c.feature_activation_num_frequencies;
        P.nht_ray_dim = P.nht_ipd * (P.nht_act == 2 ? 2 * P.nht_nf : (P.nht_act == 1 ? P.nht_nf : 1));
    }
    return P;
}

static int ensure_particle_scratch(GutHandle* h, uint32_t N) {
    const size_t n = N ? N : 1;
    GRUT_CHECK(h->tiles_count.ensure(n * 4, 1.25f));
    GRUT_CHECK(h->proj_pos.ensure(n * 8, 1.25f));
    GRUT_CHECK(h->conic_opacity.ensure(n * 16, 1.25f));
    GRUT_CHECK(h->extent.ensure(n * 8, 1.25f));
    GRUT_CHECK(h->depth.ensure(n * 4, 1.25f));
    GRUT_CHECK(h->rgb.ensure(n * 12, 1.25f));
    GRUT_CHECK(h->depth_key.ensure(n * 4, 1.25f));
    GRUT_CHECK(h->particle_idx.ensure(n * 4, 1.25f));
    GRUT_CHECK(h->depth_key_tmp.ensure(n * 4, 1.25f));
    GRUT_CHECK(h->particle_idx_tmp.ensure(n * 4, 1.25f));
    GRUT_CHECK(h->offsets.ensure(n * 4, 1.25f));
    GRUT_CHECK(h->part_offset.ensure(n * 4, 1.25f));
    GRUT_CHECK(h->rec64.ensure(n * 64, 1.25f));
    GRUT_CHECK(h->walk8.ensure(n * 8, 1.25f));
    GRUT_CHECK(h->sort_scratch.ensure(sort_scratch_bytes((uint32_t)(n * 1.25f) + 4096)));
    GRUT_CHECK(h->scan_scratch.ensure(scan_scratch_bytes((uint32_t)(n * 1.25f) + 4096)));
    if (!h->counters.ptr) {   // [1] counts the visible particles of a frame: zero once here, re-armed on the device after every read
        GRUT_CHECK(h->counters.ensure(kGutCounterWords * 4));   // 64 replicas of the counter, one 128-byte line each (gut_project_kernel)
        GRUT_HIP(hipMemset(h->counters.ptr, 0, kGutCounterWords * 4));
    }
    return GRUT_OK;
}

static int ensure_intersection_scratch(GutHandle* h, uint32_t I, uint32_t tiles) {
    const size_t n = I ? I : 1;
    if (n > h->tile_capacity) h->tile_capacity = (uint32_t)n;
    GRUT_CHECK(h->tile_keys.ensure(n * 4, 1.3f));
    GRUT_CHECK(h->tile_vals.ensure(n * 4, 1.3f));
    GRUT_CHECK(h->tile_keys_tmp.ensure(n * 4, 1.3f));
    GRUT_CHECK(h->tile_vals_tmp.ensure(n * 4, 1.3f));
    GRUT_CHECK(h->pos_particle.ensure(n * 4, 1.3f));
    GRUT_CHECK(h->tile_sort_scratch.ensure(sort_scratch_bytes((uint32_t)n), 1.3f));
    GRUT_CHECK(h->ranges.ensure((size_t)tiles * 8 + 8));
    const size_t nb = (size_t)I / kGutSegment + 1;
    // {T, C} + D per segment boundary, half tile and pixel: 80 B of space per tile entry with 64-entry segments, touched only where a forward
    // wave arrives alive.  Only the unsorted SH sweeps read them: the feature sweeps keep their own table (ck_nht below), the sorted mode
    // and the strip kernels have no checkpoints - those configurations do not pay for the space (several GB at tens of millions of entries)
    if (!h->params.nht && h->params.k_buffer == 0) {
        GRUT_CHECK(h->ck_tc.ensure(nb * 4 * 64 * 16, 1.3f));
        GRUT_CHECK(h->ck_d.ensure(nb * 4 * 64 * 4, 1.3f));
    }
    GRUT_CHECK(h->ck_reached.ensure(nb * 4, 1.3f));
    GRUT_CHECK(h->ck_boundary_tile.ensure(nb * 4, 1.3f));
    h->checkpoints.tc = h->ck_tc.as<float4>();
    h->checkpoints.d = h->ck_d.as<float>();
    h->checkpoints.reached = h->ck_reached.as<uint8_t>();
    h->checkpoints.boundary_tile = h->ck_boundary_tile.as<uint32_t>();
    h->checkpoints.num_boundaries = (uint32_t)nb;
    h->ck_boundaries_capacity = (uint32_t)nb;
    if (nht_fast_path(h->params)) GRUT_CHECK(h->ck_nht.ensure(nht_checkpoint_bytes((uint32_t)nb), 1.3f));   // {T, D, 24 partial sums} per pixel
    return GRUT_OK;
}

static GutProjected projected_view(GutHandle* h) {
    GutProjected p;
    p.tiles_count = h->tiles_count.as<uint32_t>();
    p.proj_pos = h->proj_pos.as<float2>();
    p.conic_opacity = h->conic_opacity.as<float4>();
    p.extent = h->extent.as<float2>();
    p.depth = h->depth.as<float>();
    p.rgb = h->rgb.as<float>();
    p.depth_key = h->depth_key.as<uint32_t>();
    p.particle_idx = h->particle_idx.as<uint32_t>();
    p.part_offset = h->part_offset.as<uint32_t>();
    p.rec64 = h->rec64.as<float4>();
    static const bool no_walk_cache = getenv("GRUT_GUT_NO_WALK_CACHE") != nullptr;   // (development switch: the expansion evaluates the tile test again)
    p.walk8 = (no_walk_cache || h->params.gx >= 4096 || h->params.gy >= 4096) ? nullptr : h->walk8.as<uint2>();
    return p;
}

extern "C" {

int grut_abi_version(void) { return GRUT_ABI_VERSION; }
int grut_set_allocator(GrutAllocFn alloc_fn, GrutFreeFn free_fn, void* user) {
    GRUT_REQUIRE((alloc_fn == nullptr) == (free_fn == nullptr), "grut_set_allocator: give both functions or neither");
    grut::ScratchAllocator& a = grut::scratch_allocator();
    a.alloc = alloc_fn;
    a.release = free_fn;
    a.user = user;
    return GRUT_OK;
}
const char* grut_last_error(void) { return grut::g_last_error; }

int gut_create(const GutConfig* config, GutHandle** handle) {
    GRUT_REQUIRE(config && handle, "gut_create: null argument");
    GRUT_CHECK(validate_config(*config));
    GutHandle* h = new GutHandle();
    h->cfg = *config;
    memset(&h->stats, 0, sizeof(h->stats));
    if (hipGetDevice(&h->device) != hipSuccess) {
        set_last_error("gut_create: no HIP device");
        delete h;
        return GRUT_ERR_RUNTIME;
    }
    if (hipHostMalloc(reinterpret_cast<void**>(&h->host_counters), 64, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&h->count_event, hipEventDisableTiming) != hipSuccess) {
        set_last_error("gut_create: pinned counter / event allocation failed");
        delete h;
        return GRUT_ERR_RUNTIME;
    }
    *handle = h;
    return GRUT_OK;
}

static void release_scratch(GutHandle* h) {
    DeviceBuffer* bufs[] = {&h->tiles_count, &h->proj_pos, &h->conic_opacity, &h->extent, &h->depth, &h->rgb, &h->depth_key,
                            &h->particle_idx, &h->depth_key_tmp, &h->particle_idx_tmp, &h->offsets, &h->sort_scratch,
                            &h->scan_scratch, &h->counters, &h->rec64, &h->walk8, &h->part_offset, &h->pos_particle, &h->grad_partial, &h->grad_flag,
                            &h->g_rgb, &h->poses_dev, &h->work_counters, &h->tile_keys, &h->tile_vals, &h->tile_keys_tmp,
                            &h->tile_vals_tmp, &h->tile_sort_scratch, &h->ranges, &h->ck_tc, &h->ck_d, &h->ck_nht, &h->ck_reached,
                            &h->ck_boundary_tile};
    for (DeviceBuffer* b : bufs) b->release();
}

// Hand every scratch buffer back (to hipFree or the caller's allocator); the handle stays usable, the next frame allocates afresh.
// For callers that hold a tracer across phases of very different size (a 3 M-particle evaluation between 100 k-particle jobs).
int gut_trim(GutHandle* h) {
    GRUT_REQUIRE(h, "gut_trim: null handle");
    if (h->fwd_stream) GRUT_HIP(hipStreamSynchronize(h->fwd_stream));
    else GRUT_HIP(hipDeviceSynchronize());
    release_scratch(h);
    h->tile_capacity = 0;
    h->have_forward = false;
    h->num_intersections = 0;
    h->sorted_pos = nullptr;
    h->sorted_tile_keys = nullptr;
    h->work_pending = false;
    h->ck_boundaries_capacity = 0;
    memset(&h->checkpoints, 0, sizeof(h->checkpoints));
    return GRUT_OK;
}

void gut_destroy(GutHandle* h) {
    if (!h) return;
    release_scratch(h);
    if (h->host_counters) (void)hipHostFree(h->host_counters);
    if (h->count_event) (void)hipEventDestroy(h->count_event);
    h->fwd_timer.destroy();
    h->bwd_timer.destroy();
    if (h->prof_created)
        for (int r = 0; r < GutHandle::kProfRing; ++r)
            for (int st = 0; st < GUT_NUM_STAGES; ++st) {
                (void)hipEventDestroy(h->prof_ev[r][st][0]);
                (void)hipEventDestroy(h->prof_ev[r][st][1]);
            }
    delete h;
}

int gut_forward(GutHandle* h, void* stream_, const GutFrame* frame, const float* particle_density, const void* particle_sph_,
                const float* ray_origin, const float* ray_direction, void* out_feat_density_, float* out_hit_distance,
                float* out_hit_count, int32_t* out_visibility) {
    // (fp32 buffers by default; IEEE half with particle_feature_half / feature_output_half: the kernels look at GutParams::sph_half / out_half)
    const float* particle_sph = reinterpret_cast<const float*>(particle_sph_);
    float* out_feat_density = reinterpret_cast<float*>(out_feat_density_);
    GRUT_REQUIRE(h && frame, "gut_forward: null handle/frame");
    GRUT_REQUIRE(frame->width > 0 && frame->height > 0, "gut_forward: empty image");
    GRUT_REQUIRE(ray_origin && ray_direction && out_feat_density && out_hit_distance && out_hit_count, "gut_forward: null ray/output buffer");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    ScratchStreamScope scratch_scope(s);
    h->have_forward = false;
    const uint32_t N = frame->num_particles;
    h->params = make_params(h->cfg, *frame);
    const GutParams& P = h->params;
    const uint32_t tiles = (uint32_t)(P.gx * P.gy);
    h->stats.num_particles = N;
    h->stats.num_tiles = tiles;
    h->stats.num_visible = 0;
    h->stats.num_intersections = 0;
    h->stats.key_bits = bits_for(tiles);
    h->num_intersections = 0;
    h->fwd_stream = s;
    h->fwd_density = particle_density;
    h->fwd_ray_o = ray_origin;
    h->fwd_ray_d = ray_direction;
    h->fwd_frame_id = frame->frame_id;
    if (N == 0) {  // nothing to render: outputs keep their initial values
        h->have_forward = true;
        return GRUT_OK;
    }
    GRUT_REQUIRE(particle_density && particle_sph && out_visibility, "gut_forward: null particle buffer");
    GRUT_REQUIRE(!h->cfg.feature_output_half || (!frame->out_features && !frame->out_opacity), "gut_forward: feature_output_half excludes out_features / out_opacity");
    GRUT_REQUIRE(!h->cfg.feature_transform_type || (!frame->out_features && !frame->out_opacity), "gut_forward: neural harmonic features exclude out_features / out_opacity");
    if (h->cfg.enable_kernel_timings) GRUT_CHECK(h->fwd_timer.begin(s));

    GRUT_CHECK(ensure_particle_scratch(h, N));
    // direct tile lists: tile and ordinal share the 32-bit sort key (an ordinal is below the tile count), frames beyond 65536 tiles keep
    // the position-payload lists
    static const bool legacy_lists = getenv("GRUT_GUT_LEGACY_LISTS") != nullptr;   // (development switch)
    const bool direct = h->stats.key_bits <= 16 && !legacy_lists;
    h->params.rec64 = direct ? h->rec64.as<float4>() : nullptr;
    h->params.ord_shift = h->stats.key_bits;
    h->params.sorted_keys = nullptr;
    if (frame->device_T_to_world) {  // poses stay on the device
        GRUT_CHECK(h->poses_dev.ensure(sizeof(FramePoses)));
        launch_frame_poses(s, frame->device_T_to_world, frame->device_T_to_world_end, h->poses_dev.as<FramePoses>());
        h->params.poses_dev = h->poses_dev.as<FramePoses>();
    }
    if (h->count_work) {
        // 4 counters + per-wave {lifetime, start, entries, list length} of the forward sweep + {lifetime, start} of the gradient sweep's tasks
        // (the gradient sweep runs up to 2 (I / 256 + tiles) tasks and I is not known yet: the kernel drops the records that do not fit)
        const size_t fwd_words = 16 + 4 * ((((size_t)tiles + 7) & ~(size_t)7) * 2);
        GRUT_CHECK(h->work_counters.ensure(fwd_words * 8 + 8192 + ((size_t)h->tile_capacity / kGutSegment + tiles + 64) * 2 * 16, 1.2f));
        GRUT_HIP(hipMemsetAsync(h->work_counters.ptr, 0, h->work_counters.bytes, s));
        h->params.work = h->work_counters.as<unsigned long long>();
        h->params.work_task_capacity = (uint32_t)((h->work_counters.bytes / 8 - fwd_words) / 2);
        h->work_pending = true;
    }
    const GutProjected proj = projected_view(h);
    uint32_t* d_counters = h->counters.as<uint32_t>();   // [1 + 32 r] = visible particles, replica r: zero at allocation, re-armed by the tail preparation

    const int slot = h->prof_slot % GutHandle::kProfRing;
    if (h->profile) {
        for (int st = 0; st < GUT_NUM_STAGES; ++st) h->prof_used[slot][st] = false;
        h->prof_fwd_slot = slot;
        h->prof_slot++;
    }
    // K1 projection
    GRUT_CHECK(h->stage_begin(GUT_STAGE_PROJECT, s, slot));
    launch_project(s, P, particle_density, particle_sph, proj, out_visibility, d_counters + 1);
    GRUT_CHECK(h->stage_end(GUT_STAGE_PROJECT, s, slot));
    GRUT_CHECK(h->stage_begin(GUT_STAGE_DEPTH_SORT, s, slot));
    // K2 depth sort of the particles (N keys): rank order = (depth bits, particle index)
    uint32_t *sorted_depth = nullptr, *rank_to_particle = nullptr;
    GRUT_CHECK(sort_pairs_u32(s, N, nullptr, 0, 32, proj.depth_key, proj.particle_idx, h->depth_key_tmp.as<uint32_t>(),
                              h->particle_idx_tmp.as<uint32_t>(), h->sort_scratch.ptr, h->sort_scratch.bytes, &sorted_depth,
                              &rank_to_particle, true));
    h->rank_to_particle = rank_to_particle;
    GRUT_CHECK(h->stage_end(GUT_STAGE_DEPTH_SORT, s, slot));
    GRUT_CHECK(h->stage_begin(GUT_STAGE_SCAN, s, slot));
    // K3 offsets[r] = sum_{r' <= r} tiles_count[rank_to_particle[r']]
    GRUT_CHECK(inclusive_scan_u32(s, N, proj.tiles_count, rank_to_particle, h->offsets.as<uint32_t>(), h->scan_scratch.ptr,
                                  h->scan_scratch.bytes));
    GRUT_CHECK(h->stage_end(GUT_STAGE_SCAN, s, slot));
    // I and Nv travel to the host asynchronously.  The rest of the frame (expansion, tile sort, ranges, compositing) is
    // enqueued SPECULATIVELY against the current capacity of the per-intersection scratch, with the true count read on
    // the device, so the GPU never waits for the host to wake up and launch a dozen small kernels (the reference blocks
    // here, gutRenderer.cu:313-321).  Only if the count turns out to exceed the capacity is the tail redone.
    // the only per-tile buffer: a frame with more tiles than any before must find it large enough for the speculative tail
    GRUT_CHECK(h->ranges.ensure((size_t)tiles * 8 + 8));
    launch_prepare_tail(s, h->offsets.as<uint32_t>() + (N - 1), d_counters + 1, h->host_counters, h->ranges.as<uint32_t>(), tiles * 2u,
                        h->ck_reached.as<uint32_t>(), h->ck_reached.ptr ? h->ck_boundaries_capacity : 0u);
    GRUT_HIP(hipEventRecord(h->count_event, s));
    const uint32_t tile_mask = (h->stats.key_bits >= 32) ? 0xFFFFFFFFu : ((1u << h->stats.key_bits) - 1u);
    bool first_tail = true;
    auto enqueue_tail = [&](uint32_t n, const uint32_t* n_dev) -> int {
        // K4 expansion in rank order
        GRUT_CHECK(h->stage_begin(GUT_STAGE_EXPAND, s, slot));
        launch_expand(s, P, proj, rank_to_particle, h->offsets.as<uint32_t>(), n, h->tile_keys.as<uint32_t>(),
                      direct ? h->tile_vals.as<uint32_t>() : nullptr /* legacy payload = position: generated by the sort */, h->pos_particle.as<uint32_t>());
        GRUT_CHECK(h->stage_end(GUT_STAGE_EXPAND, s, slot));
        // K5 stable radix passes over the tile bits only
        GRUT_CHECK(h->stage_begin(GUT_STAGE_TILE_SORT, s, slot));
        uint32_t *sorted_tiles = nullptr, *sorted_idx = nullptr;
        GRUT_CHECK(sort_pairs_u32(s, n, n_dev, 0, (int)h->stats.key_bits, h->tile_keys.as<uint32_t>(), h->tile_vals.as<uint32_t>(),
                                  h->tile_keys_tmp.as<uint32_t>(), h->tile_vals_tmp.as<uint32_t>(), h->tile_sort_scratch.ptr,
                                  h->tile_sort_scratch.bytes, &sorted_tiles, &sorted_idx, !direct));
        h->sorted_tile_keys = sorted_tiles;
        h->params.sorted_keys = sorted_tiles;
        h->sorted_pos = sorted_idx;
        GRUT_CHECK(h->stage_end(GUT_STAGE_TILE_SORT, s, slot));
        // K6 tile ranges
        GRUT_CHECK(h->stage_begin(GUT_STAGE_TILE_RANGES, s, slot));
        if (!first_tail) {   // (the first tail of a frame finds both tables cleared by the tail preparation)
            GRUT_HIP(hipMemsetAsync(h->ranges.ptr, 0, (size_t)tiles * 8, s));
            GRUT_HIP(hipMemsetAsync(h->ck_reached.ptr, 0, (size_t)h->ck_boundaries_capacity * 4, s));
        }
        first_tail = false;
        launch_tile_ranges(s, n, n_dev, tile_mask, tiles, sorted_tiles, h->ranges.as<uint32_t>(), h->checkpoints.boundary_tile);
        GRUT_CHECK(h->stage_end(GUT_STAGE_TILE_RANGES, s, slot));
        // K7 compositing
        GRUT_CHECK(h->stage_begin(GUT_STAGE_RENDER_FWD, s, slot));
        if (nht_fast_path(P))
            launch_render_nhtp_fwd(s, P, h->ranges.as<uint32_t>(), sorted_idx, h->pos_particle.as<uint32_t>(), particle_density, particle_sph, ray_origin,
                                   ray_direction, out_feat_density, out_hit_distance, out_hit_count, h->ck_nht.ptr, h->checkpoints, true);
        else if (P.nht)
            launch_render_nht_fwd(s, P, h->ranges.as<uint32_t>(), sorted_idx, h->pos_particle.as<uint32_t>(), particle_density, particle_sph, ray_origin,
                                  ray_direction, out_feat_density, out_hit_distance, out_hit_count);
        else if (P.k_buffer > 0)
            launch_render_k_fwd(s, P, h->ranges.as<uint32_t>(), sorted_idx, h->pos_particle.as<uint32_t>(), particle_density, proj.rgb, ray_origin,
                                ray_direction, out_feat_density, out_hit_distance, out_hit_count);
        else
            launch_render_fwd(s, P, h->ranges.as<uint32_t>(), sorted_idx, h->pos_particle.as<uint32_t>(), particle_density, proj.rgb, ray_origin,
                              ray_direction, out_feat_density, out_hit_distance, out_hit_count, h->checkpoints, true);
        GRUT_CHECK(h->stage_end(GUT_STAGE_RENDER_FWD, s, slot));
        return GRUT_OK;
    };
    const bool speculative = h->tile_capacity > 0;
    if (speculative) GRUT_CHECK(enqueue_tail(h->tile_capacity, h->offsets.as<uint32_t>() + (N - 1)));
    GRUT_HIP(hipEventSynchronize(h->count_event));
    const uint32_t I = h->host_counters[0];
    h->stats.num_visible = h->host_counters[1];
    h->stats.num_intersections = I;
    h->num_intersections = I;
    if (I == 0) {  // gutRenderer.cu:323-325: nothing is composited, the outputs are the reference's initial values
        const float far = 1e6f;
        uint32_t bits;
        memcpy(&bits, &far, 4);
        GRUT_HIP(hipMemsetAsync(out_feat_density, 0, (size_t)P.W * P.H * (P.nht ? P.nht_ray_dim + 1 : 4) * (P.out_half ? 2 : 4), s));
        GRUT_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(out_hit_distance), (int)bits, (size_t)P.W * P.H, s));
        GRUT_HIP(hipMemsetAsync(out_hit_count, 0, (size_t)P.W * P.H * 4, s));
        if (P.out_features) GRUT_HIP(hipMemsetAsync(P.out_features, 0, (size_t)P.W * P.H * 12, s));
        if (P.out_opacity) GRUT_HIP(hipMemsetAsync(P.out_opacity, 0, (size_t)P.W * P.H * 4, s));
        if (h->cfg.enable_kernel_timings) GRUT_CHECK(h->fwd_timer.end(s));
        h->have_forward = true;
        return GRUT_OK;
    }
    if (!speculative || I > h->tile_capacity) {
        GRUT_CHECK(ensure_intersection_scratch(h, I + I / 2, tiles));  // head-room: the next frames speculate against it
        first_tail = false;   // (re)allocated tables: clear them explicitly
        GRUT_CHECK(enqueue_tail(I, nullptr));
    }
    h->checkpoints.num_boundaries = I / kGutSegment + 1;  // what the gradient sweep iterates over
    GRUT_HIP(hipGetLastError());
    if (h->cfg.enable_kernel_timings) GRUT_CHECK(h->fwd_timer.end(s));
    h->have_forward = true;
    return GRUT_OK;
}

// gut_backward / gut_backward_factored: exactly one of grad_particle_sph and grad_radiance is non-null
static int backward_impl(GutHandle* h, void* stream_, const GutFrame* frame, const float* particle_density, const void* particle_sph_,
                         const float* ray_origin, const float* ray_direction, const void* feat_density_, const float* grad_feat_density,
                         const float* hit_distance, const float* grad_hit_distance, float* grad_particle_density, float* grad_particle_sph,
                         float* grad_radiance, const GutGradIO* io = nullptr, uint32_t num_chunks = 1, GrutChunkFn on_chunk = nullptr,
                         void* chunk_user = nullptr) {
    const float* particle_sph = reinterpret_cast<const float*>(particle_sph_);       // (fp32, or half with particle_feature_half)
    const float* feat_density = reinterpret_cast<const float*>(feat_density_);       // (fp32, or half with feature_output_half)
    GRUT_REQUIRE(h && frame, "gut_backward: null handle/frame");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    ScratchStreamScope scratch_scope(s);
    if (!h->have_forward || h->fwd_stream != s) {  // gutRenderer.cu:436-440
        set_last_error("gut_backward: no forward context on this stream");
        return GRUT_ERR_NOT_READY;
    }
    const GutParams& P = h->params;
    GRUT_REQUIRE(frame->num_particles == P.N && frame->width == P.W && frame->height == P.H, "gut_backward: frame differs from the forward frame");
    if (frame->frame_id != h->fwd_frame_id || (P.N > 0 && (particle_density != h->fwd_density || ray_origin != h->fwd_ray_o || ray_direction != h->fwd_ray_d))) {
        set_last_error("gut_backward: the forward context belongs to another forward (a later gut_forward ran on this handle before this "
                       "backward); run each backward before the next forward, or use one handle per in-flight frame");
        return GRUT_ERR_NOT_READY;
    }
    if (P.N == 0) return GRUT_OK;
    if (P.nht) {
        // neural harmonic features: grad_particle_density [N,12] (or the four tensors of io) and grad_particle_sph = the feature buffer's
        // gradient [N, particle_feature_dim], fp32, fully written here; grad_feat_density is [H,W,ray_dim+1] (io: [H,W,ray_dim] + [H,W,1])
        if (grad_radiance) {
            set_last_error("gut_backward_factored: neural harmonic features have no per-view factorisation; reduce the dense gradients");
            return GRUT_ERR_UNSUPPORTED;
        }
        const bool fast = nht_fast_path(P);
        if (io && !fast) {
            set_last_error("gut_backward_unpacked: this feature shape runs on the generic kernels, which write packed gradients only");
            return GRUT_ERR_UNSUPPORTED;
        }
        GRUT_REQUIRE(particle_density && particle_sph && feat_density && (io || grad_feat_density) && hit_distance && (io || grad_particle_density) &&
                         grad_particle_sph, "gut_backward: null buffer");
        if (h->cfg.enable_kernel_timings) GRUT_CHECK(h->bwd_timer.begin(s));
        GRUT_HIP(hipMemsetAsync(grad_particle_sph, 0, (size_t)P.N * P.nht_k * 4, s));
        const size_t I = h->num_intersections;
        if (fast) {
            const int slot = h->prof_fwd_slot;
            const GutProjected proj = projected_view(h);
            const GutGradOut g_out = io ? GutGradOut{nullptr, io->grad_positions, io->grad_density, io->grad_rotation, io->grad_scale}
                                        : GutGradOut{grad_particle_density, nullptr, nullptr, nullptr, nullptr};
            if (io) {
                GRUT_REQUIRE(io->grad_positions && io->grad_density && io->grad_rotation && io->grad_scale, "gut_backward_unpacked: null gradient output");
                GRUT_REQUIRE((reinterpret_cast<uintptr_t>(io->grad_rotation) & 15u) == 0, "gut_backward_unpacked: grad_rotation must be 16-byte aligned");
            }
            GutGradSlots slots;
            slots.stride = 16;
            slots.partial = nullptr;
            slots.flag = nullptr;
            slots.pos_particle = h->pos_particle.as<uint32_t>();
            if (I > 0) {
                GRUT_CHECK(h->grad_partial.ensure(2 * I * (size_t)slots.stride * 4, 1.3f));
                GRUT_CHECK(h->grad_flag.ensure(2 * I + 96, 1.3f));
                slots.partial = h->grad_partial.as<float>();
                slots.flag = h->grad_flag.as<uint8_t>();
                GRUT_HIP(hipMemsetAsync(slots.flag, 0, 2 * I, s));
                GRUT_CHECK(h->stage_begin(GUT_STAGE_RENDER_BWD, s, slot));
                launch_render_nhtp_bwd(s, P, h->ranges.as<uint32_t>(), h->sorted_pos, particle_density, particle_sph, ray_origin, ray_direction, feat_density,
                                       io ? nullptr : grad_feat_density, io ? io->grad_features : nullptr, io ? io->grad_opacity : nullptr, hit_distance,
                                       grad_hit_distance, slots, grad_particle_sph, h->ck_nht.ptr, h->checkpoints);
                GRUT_CHECK(h->stage_end(GUT_STAGE_RENDER_BWD, s, slot));
            }
            GRUT_CHECK(h->stage_begin(GUT_STAGE_PROJECT_BWD, s, slot));
            launch_grad_finalize_nht(s, P, proj, particle_density, slots, I > 0, g_out);
            GRUT_CHECK(h->stage_end(GUT_STAGE_PROJECT_BWD, s, slot));
        } else {
            GRUT_HIP(hipMemsetAsync(grad_particle_density, 0, (size_t)P.N * 48, s));
            if (I > 0)
                launch_render_nht_bwd(s, P, h->ranges.as<uint32_t>(), h->sorted_pos, h->pos_particle.as<uint32_t>(), particle_density, particle_sph, ray_origin,
                                      ray_direction, feat_density, grad_feat_density, hit_distance, grad_hit_distance, grad_particle_density, grad_particle_sph);
        }
        GRUT_HIP(hipGetLastError());
        if (h->cfg.enable_kernel_timings) GRUT_CHECK(h->bwd_timer.end(s));
        return GRUT_OK;
    }
    // the gradient tensors in the reference's packed layout, or (io) in the caller's own: see GutGradIO
    const GutGradIn g_in = io ? GutGradIn{nullptr, io->grad_features, io->grad_opacity} : GutGradIn{grad_feat_density, nullptr, nullptr};
    const GutGradOut g_out = io ? GutGradOut{nullptr, io->grad_positions, io->grad_density, io->grad_rotation, io->grad_scale}
                                : GutGradOut{grad_particle_density, nullptr, nullptr, nullptr, nullptr};
    if (io) {
        GRUT_REQUIRE(io->grad_positions && io->grad_density && io->grad_rotation && io->grad_scale, "gut_backward_unpacked: null gradient output");
        GRUT_REQUIRE((reinterpret_cast<uintptr_t>(io->grad_rotation) & 15u) == 0, "gut_backward_unpacked: grad_rotation must be 16-byte aligned");
        if (P.k_buffer > 0) {
            set_last_error("gut_backward_unpacked: the sorted (k_buffer_size > 0) backward accumulates into packed rows; use gut_backward");
            return GRUT_ERR_UNSUPPORTED;
        }
    }
    GRUT_REQUIRE(particle_density && particle_sph && feat_density && (io || grad_feat_density) && hit_distance && (io || grad_particle_density) &&
                     (grad_particle_sph || grad_radiance), "gut_backward: null buffer");  // grad_hit_distance may be NULL (no depth gradient)
    if (h->cfg.enable_kernel_timings) GRUT_CHECK(h->bwd_timer.begin(s));
    const GutProjected proj = projected_view(h);
    const int slot = h->prof_fwd_slot;
    const bool has_gdist = grad_hit_distance != nullptr;
    GutGradSlots slots;
    slots.stride = has_gdist ? 20 : 16;
    slots.partial = nullptr;
    slots.flag = nullptr;
    slots.pos_particle = h->pos_particle.as<uint32_t>();
    const size_t I = h->num_intersections;
    if (P.k_buffer > 0) {
        // sorted mode: per-hit atomics like the reference (gutKBufferRenderer.cuh:158-198), then the projection backward
        GRUT_CHECK(h->g_rgb.ensure((size_t)P.N * 12, 1.25f));
        GRUT_HIP(hipMemsetAsync(h->g_rgb.ptr, 0, (size_t)P.N * 12, s));
        GRUT_HIP(hipMemsetAsync(grad_particle_density, 0, (size_t)P.N * 48, s));
        if (I > 0) {
            GRUT_CHECK(h->stage_begin(GUT_STAGE_RENDER_BWD, s, slot));
            launch_render_k_bwd(s, P, h->ranges.as<uint32_t>(), h->sorted_pos, h->pos_particle.as<uint32_t>(), particle_density, proj.rgb, ray_origin,
                                ray_direction, feat_density, grad_feat_density, hit_distance, grad_hit_distance, grad_particle_density,
                                h->g_rgb.as<float>());
            GRUT_CHECK(h->stage_end(GUT_STAGE_RENDER_BWD, s, slot));
        }
        GRUT_CHECK(h->stage_begin(GUT_STAGE_PROJECT_BWD, s, slot));
        launch_project_bwd(s, P, proj, particle_density, particle_sph, h->g_rgb.as<float>(), g_out, grad_particle_sph, grad_radiance);
        GRUT_CHECK(h->stage_end(GUT_STAGE_PROJECT_BWD, s, slot));
        GRUT_HIP(hipGetLastError());
        if (h->cfg.enable_kernel_timings) GRUT_CHECK(h->bwd_timer.end(s));
        return GRUT_OK;
    }
    if (I > 0) {
        // one slot per (tile entry, half tile); only flagged slots are ever read, so only the flags are cleared
        GRUT_CHECK(h->grad_partial.ensure(2 * I * (size_t)slots.stride * 4, 1.3f));
        GRUT_CHECK(h->grad_flag.ensure(2 * I + 96, 1.3f));  // + slack: flags are scanned 64 at a time
        slots.partial = h->grad_partial.as<float>();
        slots.flag = h->grad_flag.as<uint8_t>();
        GRUT_HIP(hipMemsetAsync(slots.flag, 0, 2 * I, s));
        GRUT_CHECK(h->stage_begin(GUT_STAGE_RENDER_BWD, s, slot));
        launch_render_bwd(s, P, h->ranges.as<uint32_t>(), h->sorted_pos, particle_density, proj.rgb, ray_origin, ray_direction, feat_density,
                          g_in, hit_distance, grad_hit_distance, slots, h->checkpoints);
        GRUT_CHECK(h->stage_end(GUT_STAGE_RENDER_BWD, s, slot));
    }
    GRUT_CHECK(h->stage_begin(GUT_STAGE_PROJECT_BWD, s, slot));
    GRUT_CHECK(h->g_rgb.ensure((size_t)P.N * 12, 1.25f));  // per-particle radiance gradient between gather and SH backward
    if (num_chunks <= 1 || !on_chunk) {
        launch_grad_finalize(s, P, proj, particle_density, particle_sph, slots, has_gdist, I > 0, h->g_rgb.as<float>(), g_out,
                             grad_particle_sph, grad_radiance);
    } else {
        // pipelined exchange: the finalisation runs chunk by chunk over the particle range (chunks start at multiples of 128: the
        // projection backward's workgroup) and the caller is told after each chunk's launches - it issues that chunk's collectives, which
        // then run under the next chunk's kernels
        const uint32_t per = ((P.N + num_chunks - 1) / num_chunks + 127u) & ~127u;
        uint32_t c = 0;
        for (uint32_t first = 0; first < P.N; first += per, ++c) {
            const uint32_t end = first + per < P.N ? first + per : P.N;
            launch_grad_finalize(s, P, proj, particle_density, particle_sph, slots, has_gdist, I > 0, h->g_rgb.as<float>(), g_out,
                                 grad_particle_sph, grad_radiance, first, end);
            on_chunk(chunk_user, c, first, end - first);
        }
    }
    GRUT_CHECK(h->stage_end(GUT_STAGE_PROJECT_BWD, s, slot));
    GRUT_HIP(hipGetLastError());
    if (h->cfg.enable_kernel_timings) GRUT_CHECK(h->bwd_timer.end(s));
    return GRUT_OK;
}

int gut_backward(GutHandle* h, void* stream, const GutFrame* frame, const float* particle_density, const void* particle_sph,
                 const float* ray_origin, const float* ray_direction, const void* feat_density, const float* grad_feat_density,
                 const float* hit_distance, const float* grad_hit_distance, float* grad_particle_density, float* grad_particle_sph) {
    return backward_impl(h, stream, frame, particle_density, particle_sph, ray_origin, ray_direction, feat_density, grad_feat_density, hit_distance,
                         grad_hit_distance, grad_particle_density, grad_particle_sph, nullptr);
}

int gut_backward_unpacked(GutHandle* h, void* stream, const GutFrame* frame, const float* particle_density, const void* particle_sph,
                          const float* ray_origin, const float* ray_direction, const void* feat_density, const float* hit_distance,
                          const float* grad_hit_distance, const GutGradIO* io, float* grad_particle_sph) {
    GRUT_REQUIRE(io, "gut_backward_unpacked: null GutGradIO");
    return backward_impl(h, stream, frame, particle_density, particle_sph, ray_origin, ray_direction, feat_density, nullptr, hit_distance,
                         grad_hit_distance, nullptr, grad_particle_sph, nullptr, io);
}

int gut_backward_factored(GutHandle* h, void* stream, const GutFrame* frame, const float* particle_density, const void* particle_sph,
                          const float* ray_origin, const float* ray_direction, const void* feat_density, const float* grad_feat_density,
                          const float* hit_distance, const float* grad_hit_distance, float* grad_particle_density, float* grad_radiance) {
    GRUT_REQUIRE(grad_radiance, "gut_backward_factored: null buffer");
    return backward_impl(h, stream, frame, particle_density, particle_sph, ray_origin, ray_direction, feat_density, grad_feat_density, hit_distance,
                         grad_hit_distance, grad_particle_density, nullptr, grad_radiance);
}

int gut_backward_factored_chunked(GutHandle* h, void* stream, const GutFrame* frame, const float* particle_density, const void* particle_sph,
                                  const float* ray_origin, const float* ray_direction, const void* feat_density, const float* grad_feat_density,
                                  const float* hit_distance, const float* grad_hit_distance, float* grad_particle_density, float* grad_radiance,
                                  uint32_t num_chunks, GrutChunkFn on_chunk, void* user) {
    GRUT_REQUIRE(grad_radiance && on_chunk && num_chunks >= 1, "gut_backward_factored_chunked: null buffer / callback");
    GRUT_REQUIRE(!h || h->cfg.k_buffer_size == 0, "gut_backward_factored_chunked: the sorted mode finalises in one piece");
    return backward_impl(h, stream, frame, particle_density, particle_sph, ray_origin, ray_direction, feat_density, grad_feat_density, hit_distance,
                         grad_hit_distance, grad_particle_density, nullptr, grad_radiance, nullptr, num_chunks < 2 ? 2 : num_chunks, on_chunk, user);
}

int grut_sph_grad_from_views(void* stream, uint32_t num_particles, uint32_t num_views, const float* view_factors, const float* positions,
                             uint32_t position_stride, int32_t n_active_features, int32_t sph_degree, float scale, float* grad_particle_sph) {
    GRUT_REQUIRE(num_views > 0 && sph_degree >= 0 && sph_degree <= 3 && position_stride >= 3, "grut_sph_grad_from_views: bad arguments");
    if (num_particles == 0) return GRUT_OK;
    GRUT_REQUIRE(view_factors && positions && grad_particle_sph, "grut_sph_grad_from_views: null buffer");
    launch_sph_grad_from_views(reinterpret_cast<hipStream_t>(stream), num_particles, num_views, view_factors, positions, position_stride,
                               n_active_features, (sph_degree + 1) * (sph_degree + 1), scale, grad_particle_sph);
    GRUT_HIP(hipGetLastError());
    return GRUT_OK;
}

int gut_timings(GutHandle* h, float* forward_ms, float* backward_ms) {
    GRUT_REQUIRE(h, "gut_timings: null handle");
    if (forward_ms) *forward_ms = h->fwd_timer.collect();
    if (backward_ms) *backward_ms = h->bwd_timer.collect();
    return GRUT_OK;
}

int gut_profile_enable(GutHandle* h, int enable) {
    GRUT_REQUIRE(h, "gut_profile_enable: null handle");
    if (enable && h->prof_created == 0) {
        for (int r = 0; r < GutHandle::kProfRing; ++r)
            for (int st = 0; st < GUT_NUM_STAGES; ++st) {
                GRUT_HIP(hipEventCreate(&h->prof_ev[r][st][0]));
                GRUT_HIP(hipEventCreate(&h->prof_ev[r][st][1]));
                h->prof_used[r][st] = false;
            }
        h->prof_created = 1;
    }
    h->profile = enable != 0;
    h->count_work = enable >= 2;
    return GRUT_OK;
}

int gut_profile_select(GutHandle* h, uint32_t stage_mask) {
    GRUT_REQUIRE(h, "gut_profile_select: null handle");
    h->profile_mask = stage_mask;
    return GRUT_OK;
}

int gut_profile_read(GutHandle* h, float* stage_ms) {
    GRUT_REQUIRE(h && stage_ms, "gut_profile_read: null argument");
    GRUT_REQUIRE(h->prof_created, "gut_profile_read: profiling was never enabled");
    for (int st = 0; st < GUT_NUM_STAGES; ++st) {
        double sum = 0;
        int n = 0;
        for (int r = 0; r < GutHandle::kProfRing; ++r) {
            if (!h->prof_used[r][st]) continue;
            float ms = 0.f;
            GRUT_HIP(hipEventSynchronize(h->prof_ev[r][st][1]));
            GRUT_HIP(hipEventElapsedTime(&ms, h->prof_ev[r][st][0], h->prof_ev[r][st][1]));
            sum += ms;
            n++;
            h->prof_used[r][st] = false;
        }
        stage_ms[st] = n ? (float)(sum / n) : -1.f;
    }
    h->prof_slot = 0;
    return GRUT_OK;
}

int gut_stats(GutHandle* h, GutStats* stats) {
    GRUT_REQUIRE(h && stats, "gut_stats: null argument");
    if (h->work_pending && h->work_counters.ptr) {   // instrumented frame: fetch the sweeps' counters (synchronises with the frame's stream)
        std::vector<unsigned long long> all(h->work_counters.bytes / 8);
        GRUT_HIP(hipMemcpyAsync(all.data(), h->work_counters.ptr, all.size() * 8, hipMemcpyDeviceToHost, h->fwd_stream));
        GRUT_HIP(hipStreamSynchronize(h->fwd_stream));
        for (int k = 0; k < 4; ++k) h->work_host[k] = all[k];
        // the forward sweep reports per workgroup (word 18 + 4 b = accepted << 32 | evaluated): sum here
        const size_t n_fwd = (((size_t)h->params.gx * h->params.gy + 7) & ~(size_t)7) * 2;
        for (size_t b = 0; b < n_fwd && 16 + 4 * b + 3 < all.size(); ++b) {
            h->work_host[0] += all[16 + 4 * b + 2] & 0xFFFFFFFFull;
            h->work_host[1] += all[16 + 4 * b + 2] >> 32;
        }
        h->work_pending = false;
    }
    *stats = h->stats;
    stats->fwd_entries_evaluated = h->count_work ? h->work_host[0] : 0;
    stats->fwd_entries_accepted = h->count_work ? h->work_host[1] : 0;
    stats->bwd_entries_evaluated = h->count_work ? h->work_host[2] : 0;
    stats->bwd_entries_accepted = h->count_work ? h->work_host[3] : 0;
    return GRUT_OK;
}

int gut_debug_fetch_work(GutHandle* h, void* stream_, unsigned long long* out, uint64_t count) {
    GRUT_REQUIRE(h && out, "gut_debug_fetch_work: null argument");
    GRUT_REQUIRE(h->work_counters.ptr && count * 8 <= h->work_counters.bytes, "gut_debug_fetch_work: no instrumented frame / count too large");
    GRUT_HIP(hipMemcpyAsync(out, h->work_counters.ptr, count * 8, hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream_)));
    return GRUT_OK;
}

int gut_debug_fetch(GutHandle* h, void* stream_, uint32_t* tiles_count, float* proj_pos, float* conic_opacity, float* extent,
                    float* depth, float* rgb, uint32_t* sorted_particle_idx, uint32_t* tile_ranges) {
    GRUT_REQUIRE(h && h->have_forward, "gut_debug_fetch: no forward context");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    ScratchStreamScope scratch_scope(s);
    const size_t N = h->params.N;
    const size_t I = h->num_intersections;
    const size_t tiles = (size_t)h->params.gx * h->params.gy;
    const hipMemcpyKind k = hipMemcpyDeviceToDevice;
    if (N) {
        if (tiles_count) GRUT_HIP(hipMemcpyAsync(tiles_count, h->tiles_count.ptr, N * 4, k, s));
        if (proj_pos) GRUT_HIP(hipMemcpyAsync(proj_pos, h->proj_pos.ptr, N * 8, k, s));
        if (conic_opacity) GRUT_HIP(hipMemcpyAsync(conic_opacity, h->conic_opacity.ptr, N * 16, k, s));
        if (extent) GRUT_HIP(hipMemcpyAsync(extent, h->extent.ptr, N * 8, k, s));
        if (depth) GRUT_HIP(hipMemcpyAsync(depth, h->depth.ptr, N * 4, k, s));
        if (rgb) GRUT_HIP(hipMemcpyAsync(rgb, h->rgb.ptr, N * 12, k, s));
    }
    if (I) {
        if (sorted_particle_idx) launch_gather_particle_idx(s, (uint32_t)I, h->sorted_pos, h->params.rec64 ? nullptr : h->pos_particle.as<uint32_t>(), sorted_particle_idx);
        if (tile_ranges) GRUT_HIP(hipMemcpyAsync(tile_ranges, h->ranges.ptr, tiles * 8, k, s));
    }
    return GRUT_OK;
}

}  // extern "C"

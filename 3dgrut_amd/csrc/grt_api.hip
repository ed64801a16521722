// grt_api.hip — host orchestration of the 3DGRT path behind the C-ABI (include/grut_amd.h).
// Plays the role of OptixTracer (threedgrt_tracer/src/optixTracer.cpp:616-1031: buildBVH / trace / traceBwd) without
// OptiX and without libtorch: all I/O buffers belong to the caller, the handle owns the BVH and scratch.
#include <cstdlib>
#include <vector>

#include "grt_internal.hpp"

using namespace grut;

struct GrtHandle {
    GrtConfig cfg;
    int device = -1;
    uint32_t N = 0;
    bool built = false;
    hipStream_t build_stream = nullptr;
    DeviceBuffer refit_todo;   // the nodes the level-synchronous refit launches leave to grt_refit_finish_kernel
    DeviceBuffer box8;   // GRUT_PRIM_CUSTOM: the particles' exact world boxes + kernelScale^2 (grt_proxy_kernel)
    uint32_t NP = 0;     // proxies of the last build (= N; 3 N for GRUT_PRIM_TRIHEXA, 2 N for GRUT_PRIM_SPHERE)
    DeviceBuffer inst, aabb, slack, scene_enc, scene, codes, ids, codes_tmp, ids_tmp, sort_scratch, nodes,
        counters, dbg_ids, dbg_count;
    uint32_t* sorted_ids = nullptr;
    uint32_t* sorted_codes = nullptr;
    float scene_host[6] = {0, 0, 0, 0, 0, 0};
    bool scene_host_valid = false;
    EventTimer fwd_timer, bwd_timer, build_timer;
    DeviceBuffer log_pool, log_table, log_nbwd, log_state;  // forward hit log (grt_internal.hpp: GrtHitLog)
    GrtHitLog log = {nullptr, nullptr, nullptr, nullptr, 0, 0};
    bool log_valid = false;        // the last forward recorded a log ...
    int log_W = 0, log_H = 0;
    // ... of exactly this forward: the backward replays the log only if it is handed the very buffers and frame the logging
    // forward saw (two train-mode forwards may precede their backwards — gradient accumulation, multi-view losses; the
    // reference re-traverses in its backward, so any order works there).  The packed particle buffer is a fresh allocation per
    // forward that the caller keeps alive until the backward, which makes its address a per-forward token.
    const void* log_density = nullptr;
    const void* log_ray_o = nullptr;
    const void* log_ray_d = nullptr;
    GrtFrame log_frame;
    GrtLists log_lists = {nullptr, nullptr, nullptr, nullptr, nullptr};   // packet lists of the logged forward (cleared when the list scratch is rebuilt)
    bool log_matches(const GrtFrame& f, const void* density, const void* ray_o, const void* ray_d) const {
        return log_valid && log_W == f.width && log_H == f.height && log_density == density && log_ray_o == ray_o && log_ray_d == ray_d &&
               log_frame.frame_id == f.frame_id && log_frame.sph_degree == f.sph_degree && log_frame.min_transmittance == f.min_transmittance &&
               log_frame.device_ray_to_world == f.device_ray_to_world &&
               memcmp(log_frame.ray_to_world, f.ray_to_world, sizeof(f.ray_to_world)) == 0;
    }
    uint32_t* log_state_host = nullptr;  // pinned copy of {chunks used, overflow, .., [7] = a packet ran out of table columns} of the last logged forward
    uint32_t log_rounds = 48;            // columns of the chunk table (trace rounds a packet may log); doubled on demand
    hipEvent_t log_event = nullptr;
    hipEvent_t list_event = nullptr;     // the entry count of build_lists has reached the host
    bool log_event_pending = false;
    // triangle mesh of the hybrid path (grt_build_mesh_bvh): its own LBVH
    DeviceBuffer m_aabb, m_slack, m_scene_enc, m_scene, m_codes, m_ids, m_codes_tmp, m_ids_tmp, m_sort_scratch, m_nodes, m_done, m_materials;
    uint32_t* m_sorted_ids = nullptr;
    uint32_t mesh_faces = 0;
    bool mesh_built = false;
    // packet lists of the forward (GrtLists): cones, per-particle bounds and records, the binning pipeline's buffers
    DeviceBuffer l_flags, l_block_cones, l_super_cones, l_pair_cache, l_inst_rel, l_key_bits, l_counts, l_pidx, l_key_tmp, l_pidx_tmp, l_offsets, l_starts, l_bounds,
        l_scan_scratch, l_sort_scratch, l_block_keys, l_vals, l_block_keys_tmp, l_vals_tmp, l_ranges;
    uint32_t* l_host = nullptr;          // pinned: {entries, uniform-origin flag}
    uint64_t list_entries = 0;           // of the last forward (0: the BVH walk served it)
    // the backward's re-derivation kernel serves a handful of rays per frame, each for milliseconds: it runs on a stream of its own
    // next to the replay instead of behind it
    hipStream_t side_stream = nullptr;
    hipEvent_t side_fork = nullptr, side_join = nullptr;
    // The TREE of the Gaussian BVH (Morton sort, hierarchy, refit: ~25 small dependent launches, 0.35 ms at 1 M particles) is built on a stream
    // of its own behind the proxy + Morton kernels: frames with one ray origin are served by the packet lists, which need the proxies and the
    // scene box only, so the list build and the trace start while the tree is still being linked; whatever walks the tree (frames without
    // lists, the hybrid tracer, a backward without the forward's lists) makes its stream wait for `tree_done` first, and so does the next
    // build before it touches the buffers the refit reads.  GRUT_GRT_SYNC_BUILD=1: everything on the caller's stream, as before.
    hipStream_t tree_stream = nullptr;
    hipEvent_t tree_fork = nullptr, tree_done = nullptr;
    bool tree_async = false;   // the last build left its tree stages on tree_stream
    int wait_tree(hipStream_t s) {
        if (tree_async) GRUT_HIP(hipStreamWaitEvent(s, tree_done, 0));
        return GRUT_OK;
    }
    unsigned long long* dbg_bwd_sig = nullptr;   // grt_debug_backward_signature: caller DEVICE buffers the next backward fills
    uint32_t* dbg_bwd_cnt = nullptr;
    DeviceBuffer work_counters;  // instrumented launches (GRUT_GRT_COUNT=1): nodes, leaf tests, processed hits, rounds, inserts
    unsigned long long work_host[16] = {};
};

static int grt_validate(const GrtConfig& c) {
    GRUT_REQUIRE(c.particle_radiance_sph_degree >= 0 && c.particle_radiance_sph_degree <= 3, "sph degree must be in [0,3]");
    if (c.primitive_type < GRUT_PRIM_INSTANCES || c.primitive_type > GRUT_PRIM_SPHERE) {
        set_last_error("primitive_type %d: instances (0), icosahedron (1), octahedron (2), tetrahedron (3), diamond (4), custom (5), trisurfel (6), trihexa (7), sphere (8) are provided", c.primitive_type);
        return GRUT_ERR_UNSUPPORTED;
    }
    const int d = c.particle_kernel_degree;
    GRUT_REQUIRE(d == 0 || d == 1 || d == 2 || d == 3 || d == 4 || d == 5 || d == 8, "unsupported particle_kernel_degree %d", d);
    if (c.feature_transform_type != 0) {
        GRUT_REQUIRE(c.feature_transform_type == 1, "feature_transform_type %d: 0 (SH) or 1 (neural harmonic features)", c.feature_transform_type);
        GRUT_REQUIRE(!c.enable_normals, "neural harmonic features: enable_normals must be off");
        // (the feature kernels index the feature rows by the log's proxy id and blend at the volumetric intersection: closed proxies only —
        // the plugin's grt_config_from_conf refuses the same combinations)
        // round 6: trihexa and sphere too (proxy -> particle at the per-hit sites), and custom with the Slang pipeline's own candidate test
        // (particleDensityHitCustom reports the UNSIGNED distance of the maximum: RayW::absdist).  trisurfel would need the surfel branches of
        // the feature programs (the blend point on the surfel's plane): not built
        GRUT_REQUIRE(c.primitive_type != GRUT_PRIM_TRISURFEL, "neural harmonic features: primitive_type trisurfel is not provided (every other proxy is)");
        GRUT_REQUIRE(c.feature_interpolation_support == 0 || c.feature_interpolation_support == 1, "feature_interpolation_support must be 0 (centre) or 1 (tetrahedra)");
        GRUT_REQUIRE(c.feature_activation_type >= 0 && c.feature_activation_type <= 3, "feature_activation_type must be 0..3");
        const int points = c.feature_interpolation_support == 1 ? 4 : 1;
        GRUT_REQUIRE(c.interp_point_feature_dim >= 1 && c.interp_point_feature_dim <= 16 && c.particle_feature_dim == points * c.interp_point_feature_dim,
                     "particle_feature_dim %d must be %d x interp_point_feature_dim (1..16)", c.particle_feature_dim, points);
        const int nf = c.feature_activation_num_frequencies;
        const int nr = c.interp_point_feature_dim * (c.feature_activation_type == 2 ? 2 * nf : (c.feature_activation_type == 1 ? nf : 1));
        GRUT_REQUIRE(nf >= 1 && nr >= 1 && nr <= 32, "ray feature dim %d: 1..32 supported", nr);
        GRUT_REQUIRE(c.particle_feature_dim + 11 <= 64, "particle_feature_dim %d: at most 53 (one wave carries a hit's gradient words)", c.particle_feature_dim);
    }
    if (c.pipeline_type != GRUT_PIPELINE_REFERENCE) {
        GRUT_REQUIRE(c.pipeline_type == GRUT_PIPELINE_BARYCENTRIC_SURFELS, "pipeline_type %d: reference (0) or barycentricSurfels (1)", c.pipeline_type);
        // (barycentricSurfelsOptix.cu reads the hit triangle's barycentrics and the trisurfel kernel's {normal, density} rows, optixTracer.cpp:735-748)
        GRUT_REQUIRE(c.primitive_type == GRUT_PRIM_TRISURFEL, "pipeline_type barycentricSurfels: primitive_type must be trisurfel");
        GRUT_REQUIRE(c.feature_transform_type == 0 && !c.particle_feature_half && !c.feature_output_half, "pipeline_type barycentricSurfels: fp32 SH radiance only");
    }
    if (c.max_hits_per_trace != 0 && c.max_hits_per_trace != kGrtMaxHits && !(c.pipeline_type == GRUT_PIPELINE_BARYCENTRIC_SURFELS && c.max_hits_per_trace == 10)) {
        set_last_error("max_hits_per_trace=%d: the hit buffer is %d entries (PipelineParameters::MaxNumHitPerTrace)", c.max_hits_per_trace, kGrtMaxHits);
        return GRUT_ERR_UNSUPPORTED;
    }
    return GRUT_OK;
}

static GrtTraceParams trace_params(const GrtHandle* h, const GrtFrame& f) {
    GrtTraceParams P;
    memset(&P, 0, sizeof(P));
    P.degree = h->cfg.particle_kernel_degree;
    P.prim = h->cfg.primitive_type;
    P.bary = h->cfg.pipeline_type == GRUT_PIPELINE_BARYCENTRIC_SURFELS;
    P.clamping = h->cfg.particle_kernel_density_clamping;
    P.box8 = h->cfg.primitive_type == GRUT_PRIM_CUSTOM ? h->box8.as<float>() : nullptr;
    P.ncoef = (h->cfg.particle_radiance_sph_degree + 1) * (h->cfg.particle_radiance_sph_degree + 1);
    P.sph_degree = f.sph_degree < h->cfg.particle_radiance_sph_degree ? f.sph_degree : h->cfg.particle_radiance_sph_degree;
    if (P.sph_degree < 0) P.sph_degree = 0;
    P.normals = h->cfg.enable_normals;
    P.hitcounts = h->cfg.enable_hitcounts;
    P.min_response = h->cfg.particle_kernel_min_response;
    P.min_alpha = h->cfg.particle_kernel_min_alpha;  // optixTracer.cpp:925 uses 1/255
    P.max_alpha = h->cfg.particle_kernel_max_alpha;
    P.min_transmittance = f.min_transmittance;
    P.W = f.width;
    P.H = f.height;
    for (int k = 0; k < 12; ++k) P.ray_to_world[k] = f.ray_to_world[k];
    P.ray_to_world_dev = f.device_ray_to_world;
    P.sph_half = h->cfg.particle_feature_half;
    P.nht = h->cfg.feature_transform_type;
    if (P.nht) {
        const GrtConfig& c = h->cfg;
        P.nht_k = c.particle_feature_dim; P.nht_ipd = c.interp_point_feature_dim; P.nht_support = c.feature_interpolation_support;
        P.nht_act = c.feature_activation_type; P.nht_nf = c.feature_activation_num_frequencies;
        P.nht_ray_dim = P.nht_ipd * (P.nht_act == 2 ? 2 * P.nht_nf : (P.nht_act == 1 ? P.nht_nf : 1));
    }
    static const int sphere_lists = getenv("GRUT_GRT_SPHERE_LISTS") ? 1 : 0;
    P.sphere_lists = sphere_lists;
    static const float list_mark = getenv("GRUT_GRT_LIST_MARK") ? (float)atof(getenv("GRUT_GRT_LIST_MARK")) : 1.0f;
    P.list_mark = list_mark;
    static const uint32_t lane_area = getenv("GRUT_GRT_LANE_AREA") ? (uint32_t)atoi(getenv("GRUT_GRT_LANE_AREA")) : 32u;
    P.bin_lane_area = lane_area < 1u ? 1u : (lane_area > 64u ? 64u : lane_area);
    P.out_half = h->cfg.feature_output_half;
    return P;
}

static GrtBvh bvh_view(const GrtHandle* h) {
    GrtBvh b;
    b.nodes = h->nodes.as<GrtNode>();
    b.inst = h->inst.as<float>();
    b.scene = h->scene.as<float>();
    b.N = h->NP;   // leaves of the tree = proxies (three per particle for GRUT_PRIM_TRIHEXA)
    return b;
}

extern "C" {

int grt_create(const GrtConfig* config, GrtHandle** handle) {
    GRUT_REQUIRE(config && handle, "grt_create: null argument");
    GRUT_CHECK(grt_validate(*config));
    GrtHandle* h = new GrtHandle();
    h->cfg = *config;
    if (hipGetDevice(&h->device) != hipSuccess) {
        set_last_error("grt_create: no HIP device");
        delete h;
        return GRUT_ERR_RUNTIME;
    }
    if (hipHostMalloc(reinterpret_cast<void**>(&h->log_state_host), 64, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&h->log_event, hipEventDisableTiming) != hipSuccess) {
        set_last_error("grt_create: pinned buffer / event allocation failed");
        delete h;
        return GRUT_ERR_RUNTIME;
    }
    *handle = h;
    return GRUT_OK;
}

static void release_scratch(GrtHandle* h) {
    DeviceBuffer* bufs[] = {&h->refit_todo, &h->box8, &h->inst, &h->aabb, &h->slack, &h->scene_enc, &h->scene, &h->codes, &h->ids, &h->codes_tmp, &h->ids_tmp,
                            &h->sort_scratch, &h->nodes, &h->counters, &h->dbg_ids, &h->dbg_count,
                            &h->work_counters, &h->l_flags, &h->l_starts, &h->l_bounds, &h->l_pair_cache, &h->l_block_cones, &h->l_super_cones, &h->l_inst_rel, &h->l_key_bits,
                            &h->l_counts, &h->l_pidx, &h->l_key_tmp, &h->l_pidx_tmp, &h->l_offsets, &h->l_scan_scratch, &h->l_sort_scratch,
                            &h->l_block_keys, &h->l_vals, &h->l_block_keys_tmp, &h->l_vals_tmp, &h->l_ranges,
                            &h->log_pool, &h->log_table, &h->log_nbwd, &h->log_state, &h->m_aabb, &h->m_slack, &h->m_scene_enc,
                            &h->m_scene, &h->m_codes, &h->m_ids, &h->m_codes_tmp, &h->m_ids_tmp, &h->m_sort_scratch, &h->m_nodes, &h->m_done, &h->m_materials};
    for (DeviceBuffer* b : bufs) b->release();
}

// Hand every scratch buffer back (hipFree or the caller's allocator, grut_set_allocator); the handle stays usable but its BVHs are gone:
// grt_build_bvh (and grt_build_mesh_bvh) must run again before the next trace.
int grt_trim(GrtHandle* h) {
    GRUT_REQUIRE(h, "grt_trim: null handle");
    GRUT_HIP(hipDeviceSynchronize());
    release_scratch(h);
    h->built = false;
    h->tree_async = false;
    h->mesh_built = false;
    h->N = 0;
    h->NP = 0;
    h->log_valid = false;
    h->log_event_pending = false;
    h->scene_host_valid = false;
    h->list_entries = 0;
    h->log = GrtHitLog{nullptr, nullptr, nullptr, nullptr, 0, 0};
    h->log_lists = GrtLists{nullptr, nullptr, nullptr, nullptr, nullptr};
    h->sorted_codes = nullptr;
    h->sorted_ids = nullptr;
    return GRUT_OK;
}

void grt_destroy(GrtHandle* h) {
    if (!h) return;
    if (h->tree_stream) (void)hipStreamSynchronize(h->tree_stream);   // (the refit may still be reading the buffers released below)
    release_scratch(h);
    if (h->log_state_host) (void)hipHostFree(h->log_state_host);
    if (h->l_host) (void)hipHostFree(h->l_host);
    if (h->log_event) (void)hipEventDestroy(h->log_event);
    if (h->list_event) (void)hipEventDestroy(h->list_event);
    if (h->side_fork) (void)hipEventDestroy(h->side_fork);
    if (h->side_join) (void)hipEventDestroy(h->side_join);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    if (h->tree_fork) (void)hipEventDestroy(h->tree_fork);
    if (h->tree_done) (void)hipEventDestroy(h->tree_done);
    if (h->tree_stream) (void)hipStreamDestroy(h->tree_stream);
    h->fwd_timer.destroy();
    h->bwd_timer.destroy();
    h->build_timer.destroy();
    delete h;
}

// OptixTracer::buildBVH (optixTracer.cpp:616-890).  rebuild = 0 keeps the tree topology and only refits boxes
// (OPTIX_BUILD_OPERATION_UPDATE); the caller decides (threedgrt_tracer/tracer.py:198-216).
int grt_build_bvh(GrtHandle* h, void* stream_, uint32_t N, const float* positions, const float* rotations, const float* scales,
                  const float* densities, int rebuild, int allow_update) {
    (void)allow_update;
    GRUT_REQUIRE(h, "grt_build_bvh: null handle");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    ScratchStreamScope scratch_scope(s);
    if (N == 0) {
        h->N = 0;
        h->NP = 0;
        h->built = true;
        return GRUT_OK;
    }
    GRUT_REQUIRE(positions && rotations && scales && densities, "grt_build_bvh: null buffer");
    // proxies: one per particle, three (one per rhombus) for GRUT_PRIM_TRIHEXA - the tree, the hit buffers and the log are keyed by PROXY
    // (GRUT_PRIM_SPHERE: two - the ray's entry into the enclosing sphere and its exit, both offered by OptiX's built-in intersector)
    const uint32_t per = h->cfg.primitive_type == GRUT_PRIM_TRIHEXA ? 3u : (h->cfg.primitive_type == GRUT_PRIM_SPHERE ? 2u : 1u);
    GRUT_REQUIRE((uint64_t)N * per <= 0x1FFFFFFEu, "grt_build_bvh: %u particles (the hit buffers keep 29 bits of proxy index)", N);
    if (!rebuild && (!h->built || h->N != N)) rebuild = 1;  // "cannot refit GAS with a different number of gaussian" (optixTracer.cpp:629-632)
    GRUT_CHECK(h->wait_tree(s));   // the previous build's tree stages read what this build is about to overwrite (and its scratch may move)
    if (h->cfg.enable_kernel_timings) GRUT_CHECK(h->build_timer.begin(s));
    const uint32_t NP = N * per;
    const size_t n = NP;
    GRUT_CHECK(h->inst.ensure(n * 48, 1.25f));
    GRUT_CHECK(h->aabb.ensure(n * 24, 1.25f));
    GRUT_CHECK(h->slack.ensure(n * 4, 1.25f));
    if (h->cfg.primitive_type == GRUT_PRIM_CUSTOM) GRUT_CHECK(h->box8.ensure(n * 32, 1.25f));
    GRUT_CHECK(h->scene_enc.ensure(grt_scene_enc_bytes()));
    GRUT_CHECK(h->scene.ensure(64));
    GRUT_CHECK(h->codes.ensure(n * 4, 1.25f));
    GRUT_CHECK(h->ids.ensure(n * 4, 1.25f));
    GRUT_CHECK(h->codes_tmp.ensure(n * 4, 1.25f));
    GRUT_CHECK(h->ids_tmp.ensure(n * 4, 1.25f));
    GRUT_CHECK(h->sort_scratch.ensure(sort_scratch_bytes((uint32_t)(n * 1.25f) + 4096)));
    GRUT_CHECK(h->nodes.ensure(n * sizeof(GrtNode), 1.25f));
    GRUT_CHECK(h->counters.ensure(n * 4, 1.25f));

    GrtBuildParams P;
    P.N = NP;
    P.degree = h->cfg.particle_kernel_degree;
    P.prim = h->cfg.primitive_type;
    P.clamping = h->cfg.particle_kernel_density_clamping;
    P.min_response = h->cfg.particle_kernel_min_response;
    uint32_t* scene_enc = h->scene_enc.as<uint32_t>();
    grt_launch_proxies(s, P, positions, rotations, scales, densities, h->inst.as<float>(), h->aabb.as<float>(), h->slack.as<float>(), scene_enc,
                       h->cfg.primitive_type == GRUT_PRIM_CUSTOM ? h->box8.as<float>() : nullptr);
    // refit-only updates keep the sorted order of the last full build, so the code / id buffers must stay untouched
    grt_launch_morton(s, NP, h->aabb.as<float>(), scene_enc, h->scene.as<float>(), rebuild ? h->codes.as<uint32_t>() : nullptr,
                      rebuild ? h->ids.as<uint32_t>() : nullptr);
    GRUT_CHECK(h->refit_todo.ensure((n + 1) * 4, 1.25f));
    // proxies, world boxes and the scene box are final here: everything behind this line links the tree (see GrtHandle::tree_stream)
    static const bool sync_build = getenv("GRUT_GRT_SYNC_BUILD") != nullptr;
    hipStream_t ts = s;
    h->tree_async = false;
    if (!sync_build) {
        if (!h->tree_stream) {
            GRUT_HIP(hipStreamCreateWithFlags(&h->tree_stream, hipStreamNonBlocking));
            GRUT_HIP(hipEventCreateWithFlags(&h->tree_fork, hipEventDisableTiming));
            GRUT_HIP(hipEventCreateWithFlags(&h->tree_done, hipEventDisableTiming));
        }
        GRUT_HIP(hipEventRecord(h->tree_fork, s));
        GRUT_HIP(hipStreamWaitEvent(h->tree_stream, h->tree_fork, 0));
        ts = h->tree_stream;
    }
    if (rebuild) {
        uint32_t *sc = nullptr, *si = nullptr;
        GRUT_CHECK(sort_pairs_u32(ts, NP, nullptr, 0, 30, h->codes.as<uint32_t>(), h->ids.as<uint32_t>(), h->codes_tmp.as<uint32_t>(),
                                  h->ids_tmp.as<uint32_t>(), h->sort_scratch.ptr, h->sort_scratch.bytes, &sc, &si));
        h->sorted_codes = sc;
        h->sorted_ids = si;
        grt_launch_hierarchy(ts, NP, sc, si, h->nodes.as<GrtNode>());
    }
    // the refit re-derives every box from the fresh proxies; on rebuild = 0 the sorted order of the last build is reused
    GRUT_HIP(hipMemsetAsync(h->counters.ptr, 0, n, ts));   // per-node "done in pass" bytes
    grt_launch_refit(ts, NP, h->aabb.as<float>(), h->slack.as<float>(), h->nodes.as<GrtNode>(), h->counters.as<uint8_t>(), h->refit_todo.as<uint32_t>());
    GRUT_HIP(hipGetLastError());
    // (the build stage's time runs to the end of the tree, on whichever stream that is: with the tree on its own stream the stage
    // overlaps the forward's list build and trace, and the stages no longer add up to the step)
    if (h->cfg.enable_kernel_timings) GRUT_CHECK(h->build_timer.end(ts));
    if (ts != s) {
        GRUT_HIP(hipEventRecord(h->tree_done, ts));
        h->tree_async = true;
    }
    h->N = N;
    h->NP = NP;
    h->built = true;
    h->build_stream = s;
    h->scene_host_valid = false;
    return GRUT_OK;
}

// Packet lists of a frame (GrtLists, DESIGN.md "3DGRT: packet lists"): for a frame with ONE ray origin the candidates of every 8x8 ray packet
// are binned once (cones x bounding spheres, the 3DGUT pipeline) and the rounds scan windows of a sorted list instead of walking the tree.
// `lists->ranges` stays null when the frame does not qualify (rays with different origins, no particle in view) or GRUT_GRT_NO_LISTS is set.
static int build_lists(GrtHandle* h, hipStream_t s, const GrtTraceParams& P, const GrtBvh& bvh, const float* ray_origin, const float* ray_direction,
                       GrtLists* out) {
    GrtLists lists = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    h->list_entries = 0;
    h->log_lists = lists;   // the list scratch is about to be overwritten: a pending backward of an older forward walks the tree
    // (triangle-mesh proxies bin by the box of the polyhedron's vertices and start from bounding-sphere distance intervals, which the packets'
    // first tests refine to exact ones like the instance path's; GRUT_GRT_NO_MESH_LISTS=1 keeps them on the tree walk)
    // (custom primitives, round 6: binned by the particle's WORLD box - whose rays the intersection program runs for - with the 3-sigma sphere of
    // the scale frame bounding the hit distance; trihexa: every rhombus by its own flat box.  GRUT_GRT_CUSTOM_WALK / _TRIHEXA_WALK / _TRISURFEL_WALK
    // = 1 keep a primitive on the tree walk)
    if (!getenv("GRUT_GRT_NO_LISTS") && h->N > 0 && !(h->cfg.primitive_type == GRUT_PRIM_CUSTOM && getenv("GRUT_GRT_CUSTOM_WALK")) && !(h->cfg.primitive_type == GRUT_PRIM_TRIHEXA && getenv("GRUT_GRT_TRIHEXA_WALK")) && !(h->cfg.primitive_type == GRUT_PRIM_TRISURFEL && getenv("GRUT_GRT_TRISURFEL_WALK")) &&
        (h->cfg.primitive_type == GRUT_PRIM_INSTANCES || !getenv("GRUT_GRT_NO_MESH_LISTS"))) {
        const uint32_t N = h->NP, nb = grt_num_blocks(P.W, P.H), ns = grt_num_super(P.W, P.H);
        if (!h->l_host) GRUT_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->l_host), 64));
        GRUT_CHECK(h->l_flags.ensure(64));
        GRUT_CHECK(h->l_block_cones.ensure(grt_cone_table_bytes(P.W, P.H), 1.25f));   // cones, pyramids, tangent-plane tables
        GRUT_CHECK(h->l_super_cones.ensure((size_t)ns * sizeof(GrtCone), 1.25f));
        GRUT_CHECK(h->l_inst_rel.ensure((size_t)N * 64, 1.25f));
        GRUT_CHECK(h->l_pair_cache.ensure(grt_pair_cache_bytes(N), 1.25f));
        for (DeviceBuffer* b4 : {&h->l_key_bits, &h->l_counts, &h->l_pidx, &h->l_key_tmp, &h->l_pidx_tmp, &h->l_offsets, &h->l_starts})
            GRUT_CHECK(b4->ensure((size_t)N * 4, 1.25f));
        GRUT_CHECK(h->l_scan_scratch.ensure(scan_scratch_bytes((uint32_t)(N * 1.25f) + 4096)));
        uint32_t* flag = h->l_flags.as<uint32_t>();
        uint32_t* dir_len = flag + 2;
        grt_launch_list_cones(s, P, ray_origin, ray_direction, flag, dir_len, h->l_block_cones.as<GrtCone>(), h->l_super_cones.as<GrtCone>());
        grt_launch_list_count(s, P, bvh, ray_origin, flag, dir_len, h->l_block_cones.as<GrtCone>(), h->l_super_cones.as<GrtCone>(),
                              h->l_inst_rel.as<float>(), h->l_key_bits.as<uint32_t>(), h->l_counts.as<uint32_t>(), h->l_pidx.as<uint32_t>(), h->l_pair_cache.ptr);
        // particles in key order, then the offsets of their entries
        GRUT_CHECK(h->l_sort_scratch.ensure(sort_scratch_bytes((uint32_t)(N * 1.25f) + 4096)));
        uint32_t *sorted_key = nullptr, *rank_to_particle = nullptr;
        // (bits 16..31: bin_particle truncates the keys; 0xFFFFFFFF - no packet reached - still sorts last)
        GRUT_CHECK(sort_pairs_u32(s, N, nullptr, 16, 32, h->l_key_bits.as<uint32_t>(), h->l_pidx.as<uint32_t>(), h->l_key_tmp.as<uint32_t>(),
                                  h->l_pidx_tmp.as<uint32_t>(), h->l_sort_scratch.ptr, h->l_sort_scratch.bytes, &sorted_key, &rank_to_particle, true));
        GRUT_CHECK(inclusive_scan_u32(s, N, h->l_counts.as<uint32_t>(), rank_to_particle, h->l_offsets.as<uint32_t>(), h->l_scan_scratch.ptr,
                                      h->l_scan_scratch.bytes));
        // The entry count and the one-origin flag travel to the host, which sizes the per-entry buffers and enqueues the rest of the list
        // build (expansion, entry sort, ranges).  Optionally (GRUT_GRT_SPECULATE) that tail is enqueued SPECULATIVELY first, against the
        // capacity the buffers have from earlier frames and with the true count read on the device, as gut_forward does for its tail; a
        // frame whose count exceeds the capacity then sizes the buffers and runs the tail again.
        grt_launch_list_check(s, N, h->l_offsets.as<uint32_t>(), flag);   // (a wrapped 32-bit total would otherwise pass for a small one)
        GRUT_HIP(hipMemcpyAsync(&h->l_host[0], h->l_offsets.as<uint32_t>() + (N - 1), 4, hipMemcpyDeviceToHost, s));
        GRUT_HIP(hipMemcpyAsync(&h->l_host[1], flag, 8, hipMemcpyDeviceToHost, s));   // {one origin, entry count overflowed}
        if (!h->list_event) GRUT_HIP(hipEventCreateWithFlags(&h->list_event, hipEventDisableTiming));
        GRUT_HIP(hipEventRecord(h->list_event, s));
        int bits = 1;
        while ((1ull << bits) <= nb) ++bits;   // smallest b with (1 << b) > nb: the all-ones pad key never aliases a packet
        uint32_t *sorted_blocks = nullptr, *sorted_ids = nullptr;   // the payload of the sort is the particle: sorted payloads = the lists
        auto enqueue_tail = [&](uint32_t n, const uint32_t* n_dev) -> int {
            grt_launch_list_expand(s, P, bvh, ray_origin, flag, dir_len, h->l_block_cones.as<GrtCone>(), h->l_super_cones.as<GrtCone>(),
                                   rank_to_particle, h->l_offsets.as<uint32_t>(), h->l_counts.as<uint32_t>(), h->l_starts.as<uint32_t>(), n,
                                   h->l_block_keys.as<uint32_t>(), h->l_vals.as<uint32_t>(), h->l_pair_cache.ptr);
            GRUT_CHECK(sort_pairs_u32(s, n, n_dev, 0, bits, h->l_block_keys.as<uint32_t>(), h->l_vals.as<uint32_t>(), h->l_block_keys_tmp.as<uint32_t>(),
                                      h->l_vals_tmp.as<uint32_t>(), h->l_sort_scratch.ptr, h->l_sort_scratch.bytes, &sorted_blocks, &sorted_ids));
            GRUT_HIP(hipMemsetAsync(h->l_ranges.ptr, 0, (size_t)nb * 8, s));
            grt_launch_list_ranges(s, n, n_dev, nb, sorted_blocks, h->l_ranges.as<uint32_t>());
            return GRUT_OK;
        };
        // what a tail over n entries needs; false: the buffers could not be had (the frame walks the tree instead of failing)
        auto ensure_entries = [&](uint32_t n) -> bool {
            bool ok = true;
            for (DeviceBuffer* b4 : {&h->l_block_keys, &h->l_vals, &h->l_block_keys_tmp, &h->l_vals_tmp}) ok = ok && b4->ensure((size_t)n * 4, 1.3f) == GRUT_OK;
            ok = ok && h->l_ranges.ensure((size_t)nb * 8, 1.25f) == GRUT_OK;
            ok = ok && h->l_bounds.ensure((size_t)n * 8, 1.3f) == GRUT_OK;
            ok = ok && h->l_sort_scratch.ensure(sort_scratch_bytes((uint32_t)(n * 1.3f) + 4096)) == GRUT_OK;
            if (!ok) (void)hipGetLastError();
            return ok;
        };
        // capacity of the per-entry buffers as they stand (entries), 0 when one of them is missing
        auto entry_capacity = [&]() -> uint32_t {
            size_t cap = h->l_bounds.bytes / 8;
            for (const DeviceBuffer* b4 : {&h->l_block_keys, &h->l_vals, &h->l_block_keys_tmp, &h->l_vals_tmp}) cap = cap < b4->bytes / 4 ? cap : b4->bytes / 4;
            if (h->l_ranges.bytes < (size_t)nb * 8) return 0u;
            while (cap > 0 && h->l_sort_scratch.bytes < sort_scratch_bytes((uint32_t)cap)) cap = cap * 3 / 4;
            return (uint32_t)(cap < 0xFFFF0000ull ? cap : 0xFFFF0000ull);
        };
        // (measured at 1 M particles / 800x800: 9.52 ms forward with the speculative tail, 9.43 without — sorting against the capacity
        // costs more than the ~30 us the host needs to answer; the host round trip stays the default, GRUT_GRT_SPECULATE=1 switches)
        const bool speculate = getenv("GRUT_GRT_SPECULATE") != nullptr;
        const uint32_t spec_cap = speculate ? entry_capacity() : 0u;
        if (spec_cap > 0) GRUT_CHECK(enqueue_tail(spec_cap, h->l_offsets.as<uint32_t>() + (N - 1)));
        GRUT_HIP(hipEventSynchronize(h->list_event));
        const uint64_t I = h->l_host[0];
        bool usable = h->l_host[1] != 0u && h->l_host[2] == 0u && I > 0 && I < 0xFFFF0000ull;
        if (usable && !(spec_cap > 0 && I <= spec_cap)) {   // the speculative tail did not cover the frame
            usable = ensure_entries((uint32_t)I);
            if (usable) GRUT_CHECK(enqueue_tail((uint32_t)I, nullptr));
        }
        if (usable) {
            lists.ranges = h->l_ranges.as<uint32_t>();
            lists.entries = sorted_ids;
            lists.inst_rel = h->l_inst_rel.as<float>();
            lists.block_cones = h->l_block_cones.as<GrtCone>();
            lists.dir_len_enc = dir_len;
            lists.bounds = h->l_bounds.as<float2>();
            h->list_entries = I;
        }
    }
    *out = lists;
    return GRUT_OK;
}

static int grt_forward_impl(GrtHandle* h, hipStream_t s, const GrtFrame* frame, const float* particle_density, const float* particle_sph,
                            const float* ray_origin, const float* ray_direction, float* out_features, float* out_density,
                            float* out_hit_distance, float* out_normals, float* out_hits_count, int32_t* out_visibility,
                            uint32_t* dbg_ids, uint32_t* dbg_count, uint32_t dbg_cap) {
    ScratchStreamScope scratch_scope(s);
    GRUT_REQUIRE(h && frame, "grt_forward: null handle/frame");
    if (!h->built) {
        set_last_error("grt_forward: build_bvh has not been called");
        return GRUT_ERR_NOT_READY;
    }
    GRUT_REQUIRE(frame->width > 0 && frame->height > 0, "grt_forward: empty image");
    GRUT_REQUIRE(frame->num_particles == h->N, "grt_forward: %u particles but the BVH holds %u", frame->num_particles, h->N);
    GRUT_REQUIRE(ray_origin && ray_direction && out_features && out_density && out_hit_distance && out_hits_count, "grt_forward: null buffer");
    if (h->N == 0) return GRUT_OK;  // outputs keep their (zero) initial values
    GRUT_REQUIRE(particle_density && particle_sph && out_visibility, "grt_forward: null particle buffer");
    GRUT_REQUIRE(!h->cfg.enable_normals || out_normals, "grt_forward: normals enabled but no buffer");
    GrtTraceParams P = trace_params(h, *frame);
    P.dbg_cap = dbg_cap;
    if (h->cfg.enable_kernel_timings) GRUT_CHECK(h->fwd_timer.begin(s));
    // hit log for the backward (only when the caller announces one: GrtFrame::keep_hits_for_backward)
    GrtHitLog log = {nullptr, nullptr, nullptr, nullptr, 0, 0};
    h->log_valid = false;
    GRUT_REQUIRE(!(P.nht && dbg_ids), "grt_debug_forward_hits: not provided with neural harmonic features");
    // (neural harmonic features: the ray features are computed FROM the log, so every forward keeps one)
    if ((frame->keep_hits_for_backward || P.nht) && !dbg_ids && !P.bary) {
        const uint32_t blocks = div_up((uint32_t)P.W, 8) * div_up((uint32_t)P.H, 8);
        // columns of the chunk table = trace rounds a packet may log: 48 to start with, doubled when a frame reports a packet that needed more
        // (state[7]; until then such a frame counts as overflowed and its backward traverses again) - spheres and trihexa offer a particle
        // several times and make rays of 200 hits and more
        uint32_t want = blocks * 10u;  // first guess; grown from the measured use of earlier frames
        if (h->log_event_pending && hipEventQuery(h->log_event) == hipSuccess) {
            h->log_event_pending = false;
            const uint32_t used = h->log_state_host[0];
            if (used + used / 2 > want) want = used + used / 2;
            if (h->log_state_host[7] && h->log_rounds < 4096u) h->log_rounds *= 2u;
        }
        const uint32_t kMaxRounds = h->log_rounds;
        if (h->log.capacity_chunks > want) want = h->log.capacity_chunks;
        if (const char* e = getenv("GRUT_GRT_LOG_CHUNKS")) want = (uint32_t)atoi(e) > 0 ? (uint32_t)atoi(e) : 1u;  // tests: force the overflow fallback
        GRUT_CHECK(h->log_pool.ensure((size_t)want * kGrtLogSlots * 64 * 4));
        GRUT_CHECK(h->log_table.ensure((size_t)blocks * kMaxRounds * 4, 1.25f));
        GRUT_CHECK(h->log_nbwd.ensure((size_t)P.W * P.H * 4, 1.25f));
        GRUT_CHECK(h->log_state.ensure(64));
        h->log.pool = h->log_pool.as<uint32_t>();
        h->log.table = h->log_table.as<uint32_t>();
        h->log.ray_flags = h->log_nbwd.as<uint32_t>();
        h->log.state = h->log_state.as<uint32_t>();
        h->log.capacity_chunks = want;
        h->log.max_rounds = kMaxRounds;
        GRUT_HIP(hipMemsetAsync(h->log.table, 0xFF, (size_t)blocks * kMaxRounds * 4, s));
        GRUT_HIP(hipMemsetAsync(h->log.state, 0, 64, s));
        log = h->log;
        h->log_valid = true;
        h->log_W = P.W;
        h->log_H = P.H;
        h->log_density = particle_density;
        h->log_ray_o = ray_origin;
        h->log_ray_d = ray_direction;
        h->log_frame = *frame;
    }
    unsigned long long* counters = nullptr;
    if (getenv("GRUT_GRT_COUNT")) {  // development aid: work statistics of the traversal, read back by grt_stats
        const size_t wbytes = 128 + 24 * ((size_t)div_up((uint32_t)P.W, 64) * div_up((uint32_t)P.H, 64) * 64 + 512);   // + 3 words per pixel block
        GRUT_CHECK(h->work_counters.ensure(wbytes));
        counters = h->work_counters.as<unsigned long long>();
        GRUT_HIP(hipMemsetAsync(counters, 0, wbytes, s));
    }
    GrtBvh bvh = bvh_view(h);
    GrtLists lists;
    GRUT_CHECK(build_lists(h, s, P, bvh, ray_origin, ray_direction, &lists));
    // the backward's exact rounds (flagged rays) scan the same lists.  (barycentricSurfels keeps no log - it has no backward - but a train-mode
    // forward keeps its lists like every other, for grt_debug_fetch_lists)
    if (h->log_valid || (P.bary && frame->keep_hits_for_backward)) h->log_lists = lists;
    if (!lists.ranges) GRUT_CHECK(h->wait_tree(s));   // this frame walks the tree
    grt_launch_trace_fwd(s, P, bvh, particle_density, particle_sph, ray_origin, ray_direction, out_features, out_density,
                         out_hit_distance, out_normals, out_hits_count, out_visibility, dbg_ids, dbg_count, counters, log, lists);
    if (P.nht) {
        // Transmittance, hit distance, counts and visibility are the trace kernel's; the per-ray features are integrated by a pass over
        // the hit log.  The log must hold the whole frame: if the pool overflowed, grow it and trace again (a host round trip per frame
        // on this path — first version).
        for (int attempt = 0; attempt < 8; ++attempt) {
            uint32_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            GRUT_HIP(hipMemcpyAsync(st, log.state, 32, hipMemcpyDeviceToHost, s));
            GRUT_HIP(hipStreamSynchronize(s));
            if (st[1] == 0u && st[7] == 0u) break;
            GRUT_REQUIRE(attempt < 7, "grt_forward: the hit log does not fit (neural harmonic features)");
            const uint32_t blocks = div_up((uint32_t)P.W, 8) * div_up((uint32_t)P.H, 8);
            if (st[7]) {   // a packet ran out of table columns: widen the table
                h->log_rounds = h->log_rounds < 4096u ? h->log_rounds * 2u : h->log_rounds;
                GRUT_CHECK(h->log_table.ensure((size_t)blocks * h->log_rounds * 4, 1.25f));
                h->log.table = h->log_table.as<uint32_t>();
                h->log.max_rounds = h->log_rounds;
            }
            const uint32_t want = st[7] && st[0] < h->log.capacity_chunks ? h->log.capacity_chunks : h->log.capacity_chunks * 2u;
            GRUT_CHECK(h->log_pool.ensure((size_t)want * kGrtLogSlots * 64 * 4));
            h->log.pool = h->log_pool.as<uint32_t>();
            h->log.capacity_chunks = want;
            GRUT_HIP(hipMemsetAsync(h->log.table, 0xFF, (size_t)blocks * h->log.max_rounds * 4, s));
            GRUT_HIP(hipMemsetAsync(h->log.state, 0, 64, s));
            log = h->log;
            grt_launch_trace_fwd(s, P, bvh, particle_density, particle_sph, ray_origin, ray_direction, out_features, out_density,
                                 out_hit_distance, out_normals, out_hits_count, out_visibility, dbg_ids, dbg_count, counters, log, lists);
        }
        grt_launch_nht_fwd(s, P, particle_density, particle_sph, ray_origin, ray_direction, out_features, log);
    }
    if (log.pool && !h->log_event_pending) {  // how much of the pool the frame used, read lazily by a later forward
        GRUT_HIP(hipMemcpyAsync(h->log_state_host, log.state, 32, hipMemcpyDeviceToHost, s));
        GRUT_HIP(hipEventRecord(h->log_event, s));
        h->log_event_pending = true;
    }
    if (counters) {
        GRUT_HIP(hipMemcpyAsync(h->work_host, counters, 128, hipMemcpyDeviceToHost, s));
        GRUT_HIP(hipStreamSynchronize(s));
        fprintf(stderr, "[grut] grt fwd: nodes %llu, leaf tests %llu, processed %llu, rounds %llu, inserts %llu (rays %d)\n", h->work_host[0],
                h->work_host[1], h->work_host[2], h->work_host[3], h->work_host[4], frame->width * frame->height);
        fprintf(stderr, "[grut] grt fwd list batches fetched (64 entries each) %llu\n", h->work_host[12]);
        fprintf(stderr, "[grut] grt fwd leaf tests: passed %llu, distance out of range %llu, box missed %llu, beyond 3 sigma %llu; wave-level leaf visits %llu, of which ran the box test %llu, the insert chain %llu\n",
                h->work_host[5], h->work_host[6], h->work_host[7], h->work_host[8], h->work_host[9], h->work_host[10], h->work_host[11]);
        fprintf(stderr, "[grut] grt fwd phases, summed over the packets' waves (ms): scans %.1f (candidate work loops %.1f with packet lists), hit log %.1f, per-hit evaluation %.1f\n", h->work_host[13] * 1e-5,
                lists.ranges ? h->work_host[0] * 1e-5 : 0.0, h->work_host[14] * 1e-5, h->work_host[15] * 1e-5);
        if (lists.ranges) h->work_host[0] = 0;   // (grt_stats reports word 0 as node visits)
    }
    GRUT_HIP(hipGetLastError());
    if (h->cfg.enable_kernel_timings) GRUT_CHECK(h->fwd_timer.end(s));
    return GRUT_OK;
}

// (particle_sph / out_features: fp32, or IEEE half with GrtConfig::particle_feature_half / feature_output_half — the kernels look at
// GrtTraceParams::sph_half / out_half)
int grt_forward(GrtHandle* h, void* stream_, const GrtFrame* frame, const float* particle_density, const void* particle_sph,
                const float* ray_origin, const float* ray_direction, void* out_features, float* out_density, float* out_hit_distance,
                float* out_normals, float* out_hits_count, int32_t* out_visibility) {
    return grt_forward_impl(h, reinterpret_cast<hipStream_t>(stream_), frame, particle_density, reinterpret_cast<const float*>(particle_sph), ray_origin, ray_direction,
                            reinterpret_cast<float*>(out_features), out_density, out_hit_distance, out_normals, out_hits_count, out_visibility, nullptr, nullptr, 0);
}

int grt_debug_forward_hits(GrtHandle* h, void* stream_, const GrtFrame* frame, const float* particle_density, const void* particle_sph,
                           const float* ray_origin, const float* ray_direction, void* out_features, float* out_density,
                           float* out_hit_distance, float* out_normals, float* out_hits_count, int32_t* out_visibility,
                           uint32_t* hit_ids, uint32_t* hit_counts, uint32_t capacity) {
    GRUT_REQUIRE(hit_ids && hit_counts && capacity > 0, "grt_debug_forward_hits: null hit buffers");
    return grt_forward_impl(h, reinterpret_cast<hipStream_t>(stream_), frame, particle_density, reinterpret_cast<const float*>(particle_sph), ray_origin, ray_direction,
                            reinterpret_cast<float*>(out_features), out_density, out_hit_distance, out_normals, out_hits_count, out_visibility, hit_ids, hit_counts,
                            capacity);
}

int grt_backward(GrtHandle* h, void* stream_, const GrtFrame* frame, const float* particle_density, const void* particle_sph_,
                 const float* ray_origin, const float* ray_direction, const void* features_, const float* density, const float* hit_distance,
                 const float* normals, const float* grad_features, const float* grad_density, const float* grad_hit_distance,
                 const float* grad_normals, float* grad_particle_density, float* grad_particle_sph) {
    const float* particle_sph = reinterpret_cast<const float*>(particle_sph_);
    const float* features = reinterpret_cast<const float*>(features_);
    (void)normals;
    (void)grad_normals;  // the reference's backward does not propagate the normal gradient either (referenceBwdOptix.cu:103-170)
    GRUT_REQUIRE(h && frame, "grt_backward: null handle/frame");
    if (h->cfg.pipeline_type == GRUT_PIPELINE_BARYCENTRIC_SURFELS) {   // (optixTracer.cpp:311-314 would look for barycentricSurfelsBwdOptix.cu: not in the checkout)
        set_last_error("grt_backward: pipeline_type barycentricSurfels is forward only (the reference ships no backward program for it)");
        return GRUT_ERR_UNSUPPORTED;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    ScratchStreamScope scratch_scope(s);
    if (!h->built) {
        set_last_error("grt_backward: build_bvh has not been called");
        return GRUT_ERR_NOT_READY;
    }
    GRUT_REQUIRE(frame->num_particles == h->N, "grt_backward: %u particles but the BVH holds %u", frame->num_particles, h->N);
    if (h->N == 0) return GRUT_OK;
    GRUT_REQUIRE(particle_density && particle_sph && ray_origin && ray_direction && features && density && hit_distance && grad_features &&
                     grad_density && grad_particle_density && grad_particle_sph, "grt_backward: null buffer");
    GrtTraceParams P = trace_params(h, *frame);
    P.bwd_sig = h->dbg_bwd_sig;
    P.bwd_cnt = h->dbg_bwd_cnt;
    if (h->cfg.enable_kernel_timings) GRUT_CHECK(h->bwd_timer.begin(s));
    GrtHitLog log = {nullptr, nullptr, nullptr, nullptr, 0, 0};
    GrtLists lists = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // replay the hits the forward of THIS frame processed; any other backward (an older forward's, see log_matches) traverses again
    if (h->log_matches(*frame, particle_density, ray_origin, ray_direction)) {
        log = h->log;
        lists = h->log_lists;
    }
    if (log.pool && getenv("GRUT_GRT_COUNT")) {   // development aid: how many rays the forward flagged for exact backward rounds
        std::vector<uint32_t> nb((size_t)P.W * P.H);
        uint32_t st[2] = {0, 0};
        GRUT_HIP(hipMemcpyAsync(nb.data(), log.ray_flags, nb.size() * 4, hipMemcpyDeviceToHost, s));
        GRUT_HIP(hipMemcpyAsync(st, log.state, 8, hipMemcpyDeviceToHost, s));
        GRUT_HIP(hipStreamSynchronize(s));
        size_t flagged = 0;
        for (uint32_t v : nb) flagged += (v >> 31);
        fprintf(stderr, "[grut] grt bwd: %zu of %zu rays re-derive their rounds (log chunks %u, overflow %u, lists %s)\n", flagged, nb.size(), st[0], st[1],
                lists.ranges ? "yes" : "no");
    }
    if (!lists.ranges) GRUT_CHECK(h->wait_tree(s));   // a backward without the forward's lists walks the tree (before the fork: the side stream inherits the wait)
    hipStream_t rederive = s;
    // whatever happens after the fork, the caller's stream must wait for the side stream again before this call returns (the
    // re-derivation may still be adding to the gradient buffers): the guard joins on every exit path
    struct SideJoin {
        GrtHandle* h;
        hipStream_t s;
        bool armed;
        int join() {
            armed = false;
            GRUT_HIP(hipEventRecord(h->side_join, h->side_stream));
            GRUT_HIP(hipStreamWaitEvent(s, h->side_join, 0));
            return GRUT_OK;
        }
        ~SideJoin() { if (armed) (void)join(); }
    } side{h, s, false};
    if (log.pool) {   // fork: the re-derivation of the flagged rays next to the replay (both only ADD to the gradient buffers)
        if (!h->side_stream) {
            GRUT_HIP(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
            GRUT_HIP(hipEventCreateWithFlags(&h->side_fork, hipEventDisableTiming));
            GRUT_HIP(hipEventCreateWithFlags(&h->side_join, hipEventDisableTiming));
        }
        GRUT_HIP(hipEventRecord(h->side_fork, s));
        side.armed = true;
        GRUT_HIP(hipStreamWaitEvent(h->side_stream, h->side_fork, 0));
        rederive = h->side_stream;
    }
    grt_launch_trace_bwd(s, rederive, P, bvh_view(h), particle_density, particle_sph, ray_origin, ray_direction, features, density, hit_distance,
                         grad_features, grad_density, grad_hit_distance, grad_particle_density, grad_particle_sph, log, lists);
    if (log.pool) GRUT_CHECK(side.join());
    GRUT_HIP(hipGetLastError());
    if (h->cfg.enable_kernel_timings) GRUT_CHECK(h->bwd_timer.end(s));
    return GRUT_OK;
}

// HybridOptixTracer::buildMeshBVH (threedgrut_playground/include/playground/hybridTracer.h:121-122): LBVH over the triangles' boxes
// with the Gaussian builder's Morton / sort / hierarchy / refit stages; rebuild = 0 + allow_update refits the existing tree.
int grt_build_mesh_bvh(GrtHandle* h, void* stream_, uint32_t num_vertices, const float* vertices, uint32_t num_faces, const int32_t* triangles,
                       int rebuild, int allow_update) {
    GRUT_REQUIRE(h, "grt_build_mesh_bvh: null handle");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    ScratchStreamScope scratch_scope(s);
    const bool refit_only = !rebuild && allow_update && h->mesh_built && h->mesh_faces == num_faces && num_faces > 0;
    h->mesh_built = false;   // raised again only when every stage below was enqueued
    if (num_faces == 0) {
        h->mesh_faces = 0;
        h->mesh_built = true;
        return GRUT_OK;
    }
    GRUT_REQUIRE(vertices && triangles && num_vertices > 0, "grt_build_mesh_bvh: null buffer");
    const size_t n = num_faces;
    GRUT_CHECK(h->m_aabb.ensure(n * 24, 1.25f));
    GRUT_CHECK(h->m_slack.ensure(n * 4, 1.25f));
    GRUT_CHECK(h->m_scene_enc.ensure(grt_scene_enc_bytes()));
    GRUT_CHECK(h->m_scene.ensure(64));
    GRUT_CHECK(h->m_codes.ensure(n * 4, 1.25f));
    GRUT_CHECK(h->m_ids.ensure(n * 4, 1.25f));
    GRUT_CHECK(h->m_codes_tmp.ensure(n * 4, 1.25f));
    GRUT_CHECK(h->m_ids_tmp.ensure(n * 4, 1.25f));
    GRUT_CHECK(h->m_sort_scratch.ensure(sort_scratch_bytes((uint32_t)(n * 1.25f) + 4096)));
    GRUT_CHECK(h->m_nodes.ensure(n * sizeof(GrtNode), 1.25f));
    GRUT_CHECK(h->m_done.ensure(n * 4, 1.25f));
    grt_launch_mesh_aabb(s, num_faces, vertices, triangles, h->m_aabb.as<float>(), h->m_slack.as<float>(), h->m_scene_enc.as<uint32_t>());
    // (a refit keeps the sorted order, hence the child codes, of the last full build: the code / id buffers stay untouched)
    grt_launch_morton(s, num_faces, h->m_aabb.as<float>(), h->m_scene_enc.as<uint32_t>(), h->m_scene.as<float>(),
                      refit_only ? nullptr : h->m_codes.as<uint32_t>(), refit_only ? nullptr : h->m_ids.as<uint32_t>());
    if (!refit_only) {
        uint32_t *sc = nullptr, *si = nullptr;
        GRUT_CHECK(sort_pairs_u32(s, num_faces, nullptr, 0, 30, h->m_codes.as<uint32_t>(), h->m_ids.as<uint32_t>(), h->m_codes_tmp.as<uint32_t>(),
                                  h->m_ids_tmp.as<uint32_t>(), h->m_sort_scratch.ptr, h->m_sort_scratch.bytes, &sc, &si));
        grt_launch_hierarchy(s, num_faces, sc, si, h->m_nodes.as<GrtNode>());
    }
    GRUT_HIP(hipMemsetAsync(h->m_done.ptr, 0, n, s));
    grt_launch_refit(s, num_faces, h->m_aabb.as<float>(), h->m_slack.as<float>(), h->m_nodes.as<GrtNode>(), h->m_done.as<uint8_t>());
    GRUT_HIP(hipGetLastError());
    h->mesh_faces = num_faces;
    h->mesh_built = true;
    return GRUT_OK;
}

// HybridOptixTracer::traceHybrid (hybridTracer.h:129-141; playgroundKernel.cu:39-352): forward only, like the reference
int grt_trace_hybrid(GrtHandle* h, void* stream_, const GrtFrame* frame, const float* particle_density, const void* particle_sph_,
                     const float* ray_origin, const float* ray_direction, const float* ray_max_t, const GrtMesh* mesh, const GrtHybridOptions* options,
                     float* out_radiance, float* out_opacity, float* out_last_ray, uint32_t* out_bounces) {
    GRUT_REQUIRE(h && frame && mesh && options, "grt_trace_hybrid: null argument");
    const float* particle_sph = reinterpret_cast<const float*>(particle_sph_);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    ScratchStreamScope scratch_scope(s);
    if (!h->built || !h->mesh_built) {
        set_last_error("grt_trace_hybrid: build_bvh / build_mesh_bvh have not been called");
        return GRUT_ERR_NOT_READY;
    }
    GRUT_REQUIRE(frame->width > 0 && frame->height > 0 && ray_origin && ray_direction && out_radiance && out_opacity, "grt_trace_hybrid: null buffer");
    GRUT_REQUIRE(frame->num_particles == h->N, "grt_trace_hybrid: %u particles but the BVH holds %u", frame->num_particles, h->N);
    GRUT_REQUIRE(mesh->num_faces == h->mesh_faces, "grt_trace_hybrid: %u faces but the mesh BVH holds %u", mesh->num_faces, h->mesh_faces);
    GRUT_REQUIRE(h->N == 0 || (particle_density && particle_sph), "grt_trace_hybrid: null particle buffer");
    GRUT_REQUIRE(mesh->num_faces == 0 || (mesh->vertices && mesh->triangles && mesh->prim_type && mesh->refractive_index), "grt_trace_hybrid: null mesh buffer");
    GRUT_REQUIRE(!(options->playground_opts & 1u) || mesh->num_faces == 0 || mesh->vertex_normals, "grt_trace_hybrid: smooth normals need vertex_normals");
    GRUT_REQUIRE((mesh->vertex_tangents != nullptr) == (mesh->vertex_has_tangents != nullptr), "grt_trace_hybrid: vertex_tangents and vertex_has_tangents go together");
    GRUT_REQUIRE(mesh->num_materials == 0 || mesh->materials, "grt_trace_hybrid: null material table");
    const GrtTraceParams P = trace_params(h, *frame);
    GRUT_CHECK(h->wait_tree(s));   // bounced rays walk the tree
    GrtBvh bvh = bvh_view(h);
    // the material table travels with the launch (a default material stands in when the caller has none: faces that need one then
    // shade with the reference's own fallback values, tracer.py:150-170)
    GrtMaterial fallback;
    memset(&fallback, 0, sizeof(fallback));
    fallback.diffuse_factor[0] = fallback.diffuse_factor[1] = fallback.diffuse_factor[2] = fallback.diffuse_factor[3] = 1.f;
    fallback.alpha_cutoff = 0.5f;
    const uint32_t nm = mesh->num_materials ? mesh->num_materials : 1u;
    GRUT_CHECK(h->m_materials.ensure((size_t)nm * sizeof(GrtMaterial), 2.f));
    GRUT_HIP(hipMemcpyAsync(h->m_materials.ptr, mesh->num_materials ? mesh->materials : &fallback, (size_t)nm * sizeof(GrtMaterial), hipMemcpyHostToDevice, s));
    GRUT_HIP(hipStreamSynchronize(s));   // (the source is the caller's pageable host memory)
    GrtMeshView mv;
    mv.nodes = h->m_nodes.as<GrtNode>();
    mv.vertices = mesh->vertices; mv.triangles = mesh->triangles; mv.vnormals = mesh->vertex_normals; mv.vtangents = mesh->vertex_tangents;
    mv.vhas_tangents = mesh->vertex_has_tangents; mv.prim_type = mesh->prim_type; mv.mat_uv = mesh->mat_uv; mv.mat_id = mesh->mat_id;
    mv.refr = mesh->refractive_index; mv.materials = h->m_materials.as<GrtMaterial>(); mv.num_materials = nm; mv.envmap = mesh->envmap;
    mv.envmap_offset[0] = mesh->envmap_offset[0]; mv.envmap_offset[1] = mesh->envmap_offset[1];
    mv.F = mesh->num_faces;
    GrtHybridParams hp;
    hp.opts = options->playground_opts;
    hp.max_pbr_bounces = options->max_pbr_bounces;
    hp.frame_number = options->frame_number;
    if (h->cfg.enable_kernel_timings) GRUT_CHECK(h->fwd_timer.begin(s));
    GrtLists lists;   // the primary segment of every path scans the frame's packet lists when the rays share their origin
    GRUT_CHECK(build_lists(h, s, P, bvh, ray_origin, ray_direction, &lists));
    grt_launch_hybrid(s, P, bvh, mv, hp, particle_density, particle_sph, ray_origin, ray_direction, ray_max_t, out_radiance, out_opacity, out_last_ray,
                      out_bounces, lists);
    GRUT_HIP(hipGetLastError());
    if (h->cfg.enable_kernel_timings) GRUT_CHECK(h->fwd_timer.end(s));
    return GRUT_OK;
}

int grt_debug_backward_signature(GrtHandle* h, unsigned long long* ray_signature, uint32_t* ray_hit_count) {
    GRUT_REQUIRE(h && ((ray_signature != nullptr) == (ray_hit_count != nullptr)), "grt_debug_backward_signature: both buffers or none");
    h->dbg_bwd_sig = ray_signature;
    h->dbg_bwd_cnt = ray_hit_count;
    return GRUT_OK;
}

int grt_timings(GrtHandle* h, float* forward_ms, float* backward_ms, float* build_ms) {
    GRUT_REQUIRE(h, "grt_timings: null handle");
    if (forward_ms) *forward_ms = h->fwd_timer.collect();
    if (backward_ms) *backward_ms = h->bwd_timer.collect();
    if (build_ms) *build_ms = h->build_timer.collect();
    return GRUT_OK;
}

int grt_debug_fetch_work(GrtHandle* h, void* stream_, unsigned long long* out, uint64_t count) {
    GRUT_REQUIRE(h && out, "grt_debug_fetch_work: null argument");
    GRUT_REQUIRE(h->work_counters.ptr && count * 8 <= h->work_counters.bytes, "grt_debug_fetch_work: no instrumented frame / count too large");
    GRUT_HIP(hipMemcpyAsync(out, h->work_counters.ptr, count * 8, hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream_)));
    return GRUT_OK;
}

int grt_stats(GrtHandle* h, GrtStats* stats) {
    GRUT_REQUIRE(h && stats, "grt_stats: null argument");
    memset(stats, 0, sizeof(*stats));
    stats->num_particles = h->N;
    stats->num_nodes = h->NP > 1 ? h->NP - 1 : (h->NP ? 1 : 0);
    stats->nodes_visited = h->work_host[0];
    stats->candidates = h->work_host[1];
    stats->processed_hits = h->work_host[2];
    stats->list_entries = h->list_entries;
    stats->packet_tests = h->work_host[9];
    stats->list_batches = h->work_host[12];
    if (h->log_valid && h->log.pool) {   // the replay kernels' own counters: one 32-byte read-back (synchronises: a diagnostics call)
        uint32_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        GRUT_HIP(hipDeviceSynchronize());
        GRUT_HIP(hipMemcpy(st, h->log.state, 32, hipMemcpyDeviceToHost));
        stats->bwd_premise_rays = st[2];
        stats->bwd_rederived_rays = st[3];
        stats->bwd_atomic_instructions = st[4];
        stats->bwd_atomic_words = (uint64_t)st[5] * 16u;
    }
    if (h->built && h->N > 0) {
        if (!h->scene_host_valid) {  // synchronises with the build stream
            GRUT_HIP(hipMemcpyAsync(h->scene_host, h->scene.ptr, 24, hipMemcpyDeviceToHost, h->build_stream));
            GRUT_HIP(hipStreamSynchronize(h->build_stream));
            h->scene_host_valid = true;
        }
        for (int k = 0; k < 6; ++k) stats->scene_aabb[k] = h->scene_host[k];
    }
    return GRUT_OK;
}

// copies the proxy instance records (inverse maps {W rows, mu}, [N,12]) of the last build to a caller DEVICE buffer
int grt_debug_fetch_instances(GrtHandle* h, void* stream_, float* instances) {
    GRUT_REQUIRE(h && h->built && instances, "grt_debug_fetch_instances: no BVH / null buffer");
    // (GRUT_PRIM_TRIHEXA keeps three identical records per particle, one per rhombus, GRUT_PRIM_SPHERE two, one per root: the caller gets one)
    const size_t per = h->N ? h->NP / h->N : 1;
    if (h->N) GRUT_HIP(hipMemcpy2DAsync(instances, 48, h->inst.ptr, 48 * per, 48, h->N, hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream_)));
    return GRUT_OK;
}

// GRUT_PRIM_CUSTOM: the particles' world boxes + kernelScale^2 ([N,8]) of the last build to a caller DEVICE buffer
int grt_debug_fetch_custom_boxes(GrtHandle* h, void* stream_, float* box8) {
    GRUT_REQUIRE(h && h->built && box8, "grt_debug_fetch_custom_boxes: no BVH / null buffer");
    if (h->cfg.primitive_type != GRUT_PRIM_CUSTOM) { set_last_error("grt_debug_fetch_custom_boxes: primitive_type is not custom"); return GRUT_ERR_NOT_READY; }
    if (h->N) GRUT_HIP(hipMemcpyAsync(box8, h->box8.ptr, (size_t)h->N * 32, hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream_)));
    return GRUT_OK;
}

// copies the packet lists of the last forward (GrtStats::list_entries > 0) to caller DEVICE buffers: ranges [blocks, 2] u32 ([first, last) of each
// 8x8 ray packet, row-major block index) and entries [list_entries] u32 (particle ids; the top bit is the library's "refined" flag)
int grt_debug_fetch_lists(GrtHandle* h, void* stream_, uint32_t* ranges, uint32_t* entries, uint64_t entry_capacity) {
    GRUT_REQUIRE(h && ranges && entries, "grt_debug_fetch_lists: null argument");
    if (!h->log_lists.ranges || h->list_entries == 0 || h->list_entries > entry_capacity) {
        set_last_error("grt_debug_fetch_lists: no packet lists kept (a training forward of a one-origin frame keeps them) or capacity %llu < %llu entries",
                       (unsigned long long)entry_capacity, (unsigned long long)h->list_entries);
        return GRUT_ERR_NOT_READY;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    const uint32_t blocks = div_up((uint32_t)h->log_W, 8) * div_up((uint32_t)h->log_H, 8);
    GRUT_HIP(hipMemcpyAsync(ranges, h->log_lists.ranges, (size_t)blocks * 8, hipMemcpyDeviceToDevice, s));
    GRUT_HIP(hipMemcpyAsync(entries, h->log_lists.entries, (size_t)h->list_entries * 4, hipMemcpyDeviceToDevice, s));
    return GRUT_OK;
}

}  // extern "C"

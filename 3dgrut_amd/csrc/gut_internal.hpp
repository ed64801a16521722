// gut_internal.hpp — types shared by gut_kernels.hip (device) and gut_api.hip (host orchestration).
#pragma once

#include "camera.hpp"
#include "common.hpp"

namespace grut {

// kernel-argument block: GutConfig + per-frame camera/poses, flattened
struct GutParams {
    int degree;
    float min_response, min_alpha, max_alpha, min_transmittance;
    int n_active, ncoef, hitcounts;
    float ut_w0m, ut_wi, ut_w0c, ut_delta, ut_margin;
    int ut_require_all, n_rs_iter, k_buffer, global_z, rect_bounding, tight_opacity, tile_culling;
    int W, H, gx, gy;
    uint32_t N;
    GrutCamera cam;
    FramePoses poses;             // derived on the host from GutFrame::pose_start / pose_end ...
    unsigned long long* work;     // optional device counters {fwd evaluated, fwd accepted, bwd evaluated, bwd accepted} (gut_profile_enable level 2)
    // "direct" tile lists (frames of at most 65536 tiles): the tile sort's payload is the PARTICLE and the upper bits of its key carry the
    // entry's ordinal among the particle's tiles, so a list entry resolves to the particle's 64-byte record in one dependent load
    // (legacy: payload = expansion position -> pos_particle[] -> three parameter rows) and the gradient slot is part_offset + ordinal
    const float4* rec64;          // [N][4] {pos, density | quat | scale, bits(part_offset) | unclamped rgb, -}: written by the projection for visible particles; null = legacy lists
    const uint32_t* sorted_keys;  // [I] sorted tile keys (the backward reads the ordinals)
    uint32_t ord_shift;           // ordinal = key >> ord_shift
    // neural harmonic features (GutConfig::feature_transform_type 1): 0 = SH radiance
    int nht, nht_k, nht_ipd, nht_support, nht_act, nht_nf, nht_ray_dim;
    int sph_half, out_half;       // fp16 feature I/O (GutConfig::particle_feature_half / feature_output_half): the SH buffer / the [H,W,4] image are IEEE half
    uint32_t work_task_capacity;  // ... and how many {lifetime, start} records of gradient-sweep tasks fit behind the forward sweep's block
    float* out_features;          // optional contiguous copies of the radiance / opacity outputs (GutFrame::out_features / out_opacity)
    float* out_opacity;
    const FramePoses* poses_dev;  // ... or on the device from GutFrame::device_T_to_world[_end] (then this is non-null)
};
#ifdef __HIPCC__
__device__ __forceinline__ const FramePoses& frame_poses(const GutParams& P) { return P.poses_dev ? *P.poses_dev : P.poses; }
#endif
void launch_frame_poses(hipStream_t s, const float* T_start, const float* T_end, FramePoses* out);
void launch_prepare_tail(hipStream_t s, const uint32_t* last_offset, uint32_t* num_visible, uint32_t* host_counters, uint32_t* ranges_words,
                         uint32_t n_ranges_words, uint32_t* reached_words, uint32_t n_reached_words);

// per-particle products of the projection (role of GutRenderForwardContext's particle buffers, gutRenderer.cu:166-177)
struct GutProjected {
    uint32_t* tiles_count;   // [N]
    float2* proj_pos;        // [N]
    float4* conic_opacity;   // [N]
    float2* extent;          // [N]
    float* depth;            // [N]
    float* rgb;              // [N,3] unclamped radiance for the camera-centre direction
    uint32_t* depth_key;     // [N] float bits of depth, 0xFFFFFFFF when the particle touches no tile
    uint32_t* particle_idx;  // [N] identity, value array of the depth sort
    uint32_t* part_offset;   // [N] start of the particle's tile entries in expansion order (valid where tiles_count > 0)
    float4* rec64;           // [N][4] see GutParams::rec64
    uint2* walk8;            // [N] {tile box | kind << 30, keep mask} of the counting pass's one-step walks (gut_project_kernel); null = the expansion tests again
};

// Gradient partials of the compositing sweep.  Tile entries are identified by their position q in EXPANSION order
// (particle-major: particle p owns [part_offset[p], part_offset[p] + tiles_count[p])).  Each (entry, half tile) task
// that hit the particle writes its wave-reduced gradient terms to slot q*2 + half and raises the slot's flag, so
// the per-particle gather (gut_grad_finalize) sums a contiguous range: no atomics, bitwise reproducible gradients.
struct GutGradSlots {
    float* partial;              // [2 I][stride]  {B(3), d density, M(9), d radiance(3)} (+ direct scale terms, pad)
    uint8_t* flag;               // [2 I]          zeroed before every gradient sweep
    const uint32_t* pos_particle;// [I]            particle of expansion position q (0xFFFFFFFF = padding)
    int stride;                  // floats per slot: 16, or 20 when a depth gradient flows in
};

// Where the geometric particle gradient goes: the reference's packed [N,12] rows, or (packed == nullptr) the model's four
// tensors directly — positions [N,3], density [N,1], rotation [N,4], scale [N,3] — which saves the caller the unpack pass.
struct GutGradOut {
    float* packed;
    float* pos;
    float* dns;
    float* rot;
    float* scl;
};
// Upstream image gradient: one [H,W,4] tensor like the reference, or (fd == nullptr) separate [H,W,3] / [H,W,1] tensors as
// autograd delivers them (either may be null = zero), which saves the caller a concatenation.
struct GutGradIn {
    const float* fd;
    const float* rgb;
    const float* opa;
};
#ifdef __HIPCC__
__device__ __forceinline__ float4 load_grad_in(const GutGradIn& g, size_t pix) {
    if (g.fd) return reinterpret_cast<const float4*>(g.fd)[pix];
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.rgb) { r.x = g.rgb[3 * pix]; r.y = g.rgb[3 * pix + 1]; r.z = g.rgb[3 * pix + 2]; }
    if (g.opa) r.w = g.opa[pix];
    return r;
}
#endif

// the gradient sweep runs a long tile list as independent segments of this many sorted entries
#ifndef GRUT_GUT_SEGMENT
#define GRUT_GUT_SEGMENT 64    // round 4: 256 -> 64 (the forward's batch): gradient-sweep tasks a quarter as long - 1080p step 2.040 -> 2.016 ms, 800x800
                               // 1.84 -> 1.75, 400x400 0.81 -> 0.70, 3 M Gaussians 3.25 -> 3.17 (A/B on one box; 128: in between); 80 B of
                               // checkpoint space per tile entry
#endif
constexpr uint32_t kGutSegment = GRUT_GUT_SEGMENT;

// Per-pixel compositing state saved by the forward sweep every kGutSegment sorted entries (at the global sorted
// index b * kGutSegment, for the tile whose list contains it), so that the gradient sweep can start any segment of
// a long tile list from it: 5 floats per lane = {T, Cr, Cg, Cb} + {D}.  `reached` marks (boundary, strip) pairs the
// forward actually crossed with a live ray.
struct GutCheckpoints {
    float4* tc;          // [num_boundaries][4 strips][64 lanes]
    float* d;            // [num_boundaries][4 strips][64 lanes]
    uint8_t* reached;    // [num_boundaries][4 strips]
    uint32_t* boundary_tile;  // [num_boundaries]
    uint32_t num_boundaries;  // boundaries b = 0 .. num_boundaries-1 at sorted index b * kGutSegment (b = 0 unused)
};

// the visible-particle counter is replicated: 15.6 k waves adding to ONE word serialise in the L2 (measured: 0.13 of the projection's
// 0.21 ms); a wave adds to replica (workgroup mod 64), each on its own 128-byte line, the tail preparation sums and re-arms them
constexpr uint32_t kGutCounterReplicas = 64, kGutCounterStride = 32, kGutCounterWords = kGutCounterReplicas * kGutCounterStride + 32;
void launch_project(hipStream_t s, const GutParams& P, const float* density12, const float* sph, const GutProjected& out,
                    int32_t* visibility, uint32_t* num_visible);
void launch_expand(hipStream_t s, const GutParams& P, const GutProjected& proj, const uint32_t* rank_to_particle,
                   const uint32_t* offsets, uint32_t capacity, uint32_t* tile_keys, uint32_t* tile_vals, uint32_t* pos_particle);
void launch_gather_particle_idx(hipStream_t s, uint32_t n, const uint32_t* sorted_pos, const uint32_t* pos_particle, uint32_t* out);
void launch_tile_ranges(hipStream_t s, uint32_t n, const uint32_t* n_dev, uint32_t tile_mask, uint32_t num_tiles,
                        const uint32_t* sorted_tile_keys, uint32_t* ranges, uint32_t* boundary_tile);
void launch_render_fwd(hipStream_t s, const GutParams& P, const uint32_t* ranges, const uint32_t* sorted_pos, const uint32_t* pos_particle,
                       const float* density12,
                       const float* rgb, const float* ray_o, const float* ray_d, float* out_fd, float* out_dist, float* out_cnt,
                       const GutCheckpoints& ck, bool write_checkpoints);
void launch_render_bwd(hipStream_t s, const GutParams& P, const uint32_t* ranges, const uint32_t* sorted_pos, const float* density12,
                       const float* rgb, const float* ray_o, const float* ray_d, const float* fd, const GutGradIn& g_fd, const float* dist,
                       const float* g_dist, const GutGradSlots& slots, const GutCheckpoints& ck);
void launch_render_nht_fwd(hipStream_t s, const GutParams& P, const uint32_t* ranges, const uint32_t* sorted_pos, const uint32_t* pos_particle,
                           const float* density12, const float* features, const float* ray_o, const float* ray_d, float* out_fd, float* out_dist,
                           float* out_cnt);
void launch_render_nht_bwd(hipStream_t s, const GutParams& P, const uint32_t* ranges, const uint32_t* sorted_pos, const uint32_t* pos_particle,
                           const float* density12, const float* features, const float* ray_o, const float* ray_d, const float* fd, const float* g_fd,
                           const float* dist, const float* g_dist, float* g_density12, float* g_features);
// the pixel-pair sweeps for the reference's default feature model (48 = 4 x 12 floats, sincos, one frequency; gut_render_nht.inl)
bool nht_fast_path(const GutParams& P);
uint64_t nht_checkpoint_bytes(uint32_t num_boundaries);
void launch_render_nhtp_fwd(hipStream_t s, const GutParams& P, const uint32_t* ranges, const uint32_t* sorted_pos, const uint32_t* pos_particle,
                            const float* density12, const float* features, const float* ray_o, const float* ray_d, float* out_fd, float* out_dist,
                            float* out_cnt, void* ck_nht, const GutCheckpoints& ck, bool write_checkpoints);
void launch_render_nhtp_bwd(hipStream_t s, const GutParams& P, const uint32_t* ranges, const uint32_t* sorted_pos, const float* density12,
                            const float* features, const float* ray_o, const float* ray_d, const float* fd, const float* g_fd, const float* g_feat,
                            const float* g_opa, const float* dist, const float* g_dist, const GutGradSlots& slots, float* g_features,
                            const void* ck_nht, const GutCheckpoints& ck);
void launch_grad_finalize_nht(hipStream_t s, const GutParams& P, const GutProjected& proj, const float* density12, const GutGradSlots& slots,
                              bool have_partials, const GutGradOut& g_out);
void launch_render_k_fwd(hipStream_t s, const GutParams& P, const uint32_t* ranges, const uint32_t* sorted_pos, const uint32_t* pos_particle,
                         const float* density12, const float* rgb, const float* ray_o, const float* ray_d, float* out_fd, float* out_dist,
                         float* out_cnt);
void launch_render_k_bwd(hipStream_t s, const GutParams& P, const uint32_t* ranges, const uint32_t* sorted_pos, const uint32_t* pos_particle,
                         const float* density12, const float* rgb, const float* ray_o, const float* ray_d, const float* fd, const float* g_fd,
                         const float* dist, const float* g_dist, float* g_density12, float* g_rgb);
void launch_project_bwd(hipStream_t s, const GutParams& P, const GutProjected& proj, const float* density12, const float* sph,
                        const float* g_rgb, const GutGradOut& g_out, float* g_sph, float* g_radiance, uint32_t first = 0u, uint32_t end = 0xFFFFFFFFu);
void launch_grad_finalize(hipStream_t s, const GutParams& P, const GutProjected& proj, const float* density12, const float* sph,
                          const GutGradSlots& slots, bool has_gdist, bool have_partials, float* g_rgb, const GutGradOut& g_out, float* g_sph,
                          float* g_radiance, uint32_t first = 0u, uint32_t end = 0xFFFFFFFFu);
void launch_sph_grad_from_views(hipStream_t s, uint32_t N, uint32_t n_views, const float* factors, const float* positions, uint32_t pos_stride,
                                int n_active, int ncoef, float scale, float* g_sph);

}  // namespace grut

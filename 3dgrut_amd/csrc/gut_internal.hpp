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
    FramePoses poses;
};

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
};

// the gradient sweep runs a long tile list as independent segments of this many sorted entries
constexpr uint32_t kGutSegment = 256;

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

void launch_project(hipStream_t s, const GutParams& P, const float* density12, const float* sph, const GutProjected& out,
                    int32_t* visibility, uint32_t* num_visible);
void launch_expand(hipStream_t s, const GutParams& P, const GutProjected& proj, const uint32_t* rank_to_particle,
                   const uint32_t* offsets, uint32_t capacity, uint32_t* tile_keys, uint32_t* tile_vals);
void launch_tile_ranges(hipStream_t s, uint32_t n, uint32_t tile_mask, uint32_t num_tiles, const uint32_t* sorted_tile_keys,
                        uint32_t* ranges, uint32_t* boundary_tile);
void launch_render_fwd(hipStream_t s, const GutParams& P, const uint32_t* ranges, const uint32_t* sorted_idx, const float* density12,
                       const float* rgb, const float* ray_o, const float* ray_d, float* out_fd, float* out_dist, float* out_cnt,
                       const GutCheckpoints& ck, bool write_checkpoints);
void launch_render_bwd(hipStream_t s, const GutParams& P, const uint32_t* ranges, const uint32_t* sorted_idx, const float* density12,
                       const float* rgb, const float* ray_o, const float* ray_d, const float* fd, const float* g_fd, const float* dist,
                       const float* g_dist, float* g_density12, float* g_rgb, const GutCheckpoints& ck);
void launch_project_bwd(hipStream_t s, const GutParams& P, const uint32_t* tiles_count, const float* density12, const float* sph,
                        const float* rgb, const float* g_rgb, float* g_density12, float* g_sph);

}  // namespace grut

// grt_internal.hpp — types shared by grt_kernels.hip (device) and grt_api.hip (host orchestration) of the 3DGRT path.
#pragma once

#include "common.hpp"

namespace grut {

// Binary BVH node, 64 bytes: the boxes of BOTH children live in the parent, so one 64-byte fetch serves two box tests.
// The two children are interleaved component by component ({lo0.x, lo1.x}, {lo0.y, lo1.y}, ...): each pair lands in two
// neighbouring registers, which is what the packed fp32 instructions of the slab test want (one instruction, both children).
// child code: bit 31 set -> leaf, low bits = particle index; kGrtNoChild = empty slot (only in a one-particle tree).
// slack = sqrt(2) * (largest proxy half axis below the child): lower bound of a candidate's hit distance is
// (box entry distance - slack), see DESIGN.md "3DGRT traversal".
struct GrtNode {
    float lox[2], loy[2], loz[2];
    float hix[2], hiy[2], hiz[2];
    uint32_t c[2];
    float slack[2];
};
static_assert(sizeof(GrtNode) == 64, "GrtNode must be 64 bytes");
constexpr uint32_t kGrtLeafBit = 0x80000000u;
constexpr uint32_t kGrtNoChild = 0xFFFFFFFFu;
constexpr int kGrtMaxHits = 16;       // PipelineParameters::MaxNumHitPerTrace (pipelineParameters.h:83)
constexpr int kGrtGather = 16;        // forward: candidates per traversal (16 = one trace round, 32 = two rounds from one walk)
constexpr int kGrtStackDepth = 64;    // a radix tree over 30+32-bit keys is at most 62 levels deep

struct GrtBuildParams {
    uint32_t N;
    int degree, clamping;
    float min_response;
};

struct GrtBvh {
    const GrtNode* nodes;    // [max(N-1, 1)] internal nodes, root = 0
    const float* inst;       // [N,12] inverse instance map {W rows, mu}: o' = W (o - mu), d' = W d
    const float* scene;      // [6] scene AABB
    uint32_t N;
};

struct GrtTraceParams {
    int degree, sph_degree, ncoef, normals, hitcounts;
    float min_response, min_alpha, max_alpha, min_transmittance;
    int W, H;
    float ray_to_world[12];
    uint32_t dbg_cap;
};

// Log of the forward's processed hits, so that the backward replays them instead of traversing again.  One chunk =
// the 16 x 64 particle ids a wave processed in one trace round ([slot][lane], 0xFFFFFFFF = not processed); a wave's
// chunks are listed in `table[block][round]`.  `nbwd[ray]` = how many of a ray's processed hits the backward visits
// (those with t < endT, referenceBwdOptix.cu:126-131).  If the pool or the table overflows, `state[1]` is raised and
// the backward falls back to traversal.
struct GrtHitLog {
    uint32_t* pool;      // [capacity_chunks][2][16][64]: particle ids, then the ray's entry distance into each proxy box
    uint32_t* table;     // [num_blocks][max_rounds]
    uint32_t* nbwd;      // [W*H]
    uint32_t* state;     // [0] = chunks allocated, [1] = overflow flag
    uint32_t capacity_chunks, max_rounds;
};

// triangle mesh of the hybrid path (playground): its own LBVH over triangle boxes + the per-face / per-vertex attributes
struct GrtMeshView {
    const GrtNode* nodes;        // [max(F-1, 1)]
    const float* vertices;       // [V,3]
    const int32_t* triangles;    // [F,3]
    const float* vnormals;       // [V,3]
    const int32_t* prim_type;    // [F] PlaygroundPrimitiveTypes
    const float* refr;           // [F]
    const float* diffuse;        // [F,3]
    uint32_t F;
};
struct GrtHybridParams {
    uint32_t opts;               // PlaygroundRenderOptions: bit 0 smooth normals, bit 1 Gaussian tracing off
    uint32_t max_pbr_bounces;
    float background[3];
};

// build stages
void grt_launch_proxies(hipStream_t s, const GrtBuildParams& P, const float* pos, const float* rot, const float* scl, const float* dns,
                        float* inst, float* aabb, float* slack, uint32_t* scene_enc);
void grt_launch_morton(hipStream_t s, uint32_t N, const float* aabb, const uint32_t* scene_enc, float* scene, uint32_t* codes, uint32_t* ids);
void grt_launch_hierarchy(hipStream_t s, uint32_t N, const uint32_t* sorted_codes, const uint32_t* sorted_ids, GrtNode* nodes);
void grt_launch_refit(hipStream_t s, uint32_t N, const float* aabb, const float* slack, GrtNode* nodes, uint8_t* done);
size_t grt_scene_enc_bytes();
// trace
void grt_launch_trace_fwd(hipStream_t s, const GrtTraceParams& P, const GrtBvh& bvh, const float* density12, const float* sph,
                          const float* ray_o, const float* ray_d, float* out_rad, float* out_dns, float* out_hit2, float* out_nrm,
                          float* out_cnt, int32_t* visibility, uint32_t* dbg_ids, uint32_t* dbg_count, unsigned long long* counters,
                          const GrtHitLog& log);
void grt_launch_trace_bwd(hipStream_t s, const GrtTraceParams& P, const GrtBvh& bvh, const float* density12, const float* sph,
                          const float* ray_o, const float* ray_d, const float* rad, const float* dns, const float* hit2, const float* g_rad,
                          const float* g_dns, const float* g_hit, float* g_density12, float* g_sph, const GrtHitLog& log);

void grt_launch_mesh_aabb(hipStream_t s, uint32_t F, const float* vertices, const int32_t* triangles, float* aabb, float* slack, uint32_t* scene_enc);
void grt_launch_hybrid(hipStream_t s, const GrtTraceParams& P, const GrtBvh& bvh, const GrtMeshView& mesh, const GrtHybridParams& hp,
                       const float* density12, const float* sph, const float* ray_o, const float* ray_d, const float* ray_max_t, float* out_rgb,
                       float* out_alpha, float* out_last_ray, uint32_t* out_bounces);

}  // namespace grut

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
constexpr int kGrtMaxDepth = 64;      // a radix tree over 30+32-bit keys is at most 62 levels deep
constexpr int kGrtStackDepth = 3 * kGrtMaxDepth;   // the wide walk (trace_round4) parks up to 3 nodes per level

struct GrtBuildParams {
    uint32_t N;
    int degree, clamping;
    float min_response;
    int prim;   // GrtConfig::primitive_type: the proxies' world boxes enclose the polyhedron, not the unit cube
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
    int prim;                 // GrtConfig::primitive_type (GRUT_PRIM_*): which candidate test a (ray, particle) pair takes
    int bary;                 // GrtConfig::pipeline_type == GRUT_PIPELINE_BARYCENTRIC_SURFELS: ten hits per trace, response from the plane crossing (forward only)
    int clamping;             // GrtConfig::particle_kernel_density_clamping (the surfel pipeline's scaled response needs it per hit)
    const float* box8;        // GRUT_PRIM_CUSTOM: [N,8] {world box, kernelScale^2, 0} of the proxy kernel (null otherwise)
    float ray_to_world[12];
    const float* ray_to_world_dev;   // optional: the same matrix in device memory (GrtFrame::device_ray_to_world), used instead when set
    uint32_t dbg_cap;
    // optional (grt_debug_backward_signature): per ray, how many hits the backward differentiated and an order-independent
    // signature of which particles they were — the parity tests compare the replayed backward with the re-derived one ray by ray
    int sph_half, out_half;   // fp16 feature I/O (GrtConfig::particle_feature_half / feature_output_half)
    int nht, nht_k, nht_ipd, nht_support, nht_act, nht_nf, nht_ray_dim;   // neural harmonic features (GrtConfig::feature_transform_type 1)
    int sphere_lists;         // development switch (GRUT_GRT_SPHERE_LISTS=1): bin by the proxies' bounding spheres only
    uint32_t bin_lane_area;   // binning: a particle with at most this many candidate cells is tested by its own lane (development switch GRUT_GRT_LANE_AREA, <= 64)
    float list_mark;          // list_round's first pass reaches list_mark x the previous round's span (development switch GRUT_GRT_LIST_MARK, default 1: measured 0.5 / 0.8 / 1 / 1.25 / 2 / off = 10.8 / 10.0 / 9.36 / 9.45 / 9.53 / 9.75 ms forward)
    unsigned long long* bwd_sig;
    uint32_t* bwd_cnt;
};

// Candidate lists of the forward for frames whose rays all start at ONE point (every pinhole / fisheye camera) — see "3DGRT: packet
// lists" in DESIGN.md.  Each 8x8-pixel ray packet (one wave) gets the list of the particles whose proxy can be touched by one of its rays
// (the packet's bounding cone against the proxy's bounding sphere), ascending in `key`, a lower bound of the hit distance of that
// particle for ANY ray of the frame; `ub` is the matching upper bound.  A k = 16 trace round is then a scan of a WINDOW of the list
// (entries whose [key, ub] can still hold a hit beyond the ray's last one, up to the 16th-nearest distance found) instead of a tree
// walk.  Built per frame with the 3DGUT binning pipeline (count / depth sort / scan / expand / stable sort by packet / ranges).
struct GrtCone;
struct GrtLists {
    const uint32_t* ranges;          // [blocks][2]: entries [first, last) of packet b (row-major 8x8 block index) in `entries`
    const uint32_t* entries;         // [I] particle of each entry; a packet's entries ascend in the particles' sort key
    const float* inst_rel;           // [N,16], one 64-byte line per particle: {W rows, W (o - mu)} (the proxy-frame ray origin is the same for
                                     // all rays, computed once) + {proxy centre - ray origin, sort key} (the key is a lower bound of the hit
                                     // distance for every ray)
    const GrtCone* block_cones;      // [blocks] bounding cone of each packet's rays (list_round derives packet-specific bounds from it)
    const uint32_t* dir_len_enc;     // [2] float bits of the smallest / largest ray direction length of the frame
    float2* bounds;                  // [I] per entry: the smallest / largest hit distance over the packet's rays that meet the proxy,
                                     // written by the forward the first time the packet tests the entry, which then sets
                                     // kGrtEntryRefined in the entry's word (entries without the bit: geometric bounds apply)
};
constexpr uint32_t kGrtEntryRefined = 0x80000000u;   // (particle indices stay below 2^31; the pad entry is all ones)
struct GrtCone {   // bounding cone of the rays of an 8x8 packet / of a 64x64-pixel super tile, apex at the common ray origin
    float ax, ay, az, cos_t, sin_t, valid, pad0, pad1;
};
// Bounding pyramid of a packet's rays, apex at the common origin: in the frame (u, w, cone axis a) every ray direction d has
// x0 <= (d.u)/(d.a) <= x1 and y0 <= (d.w)/(d.a) <= y1 (u follows the packet's pixel rows, so for a pinhole grid the pyramid is the
// packet's own frustum, where the cone circumscribes it).  The binning tests the proxy BOX against its four side planes and the
// cone's tangent plane; [blocks] of them follow the [blocks] cones in the same allocation (grt_block_pyramids).  ok = 0: no pyramid
// (a packet whose cone is 90 degrees or wider), only the cone test applies.
struct GrtPyramid {
    float ux, uy, uz, x0, wx, wy, wz, x1, y0, y1, ok, pad;
};
__host__ __device__ inline const GrtPyramid* grt_block_pyramids(const GrtCone* block_cones, uint32_t num_blocks) {
    return reinterpret_cast<const GrtPyramid*>(block_cones + num_blocks);
}
// The frame's tangent plane, behind the pyramids in the same allocation (grt_cone_table_bytes): one frame (u, w, a) for ALL rays —
// a = the centre pixel's direction, u along the pixel rows — in which packet b's rays have x0 <= (d.u)/(d.a) <= x1, y0 <= (d.w)/(d.a)
// <= y1 (`rects[b]`), packet column c lies within cols[c] and packet row r within rows[r] (unions over the column / the row: for a
// pinhole grid they ARE the column's and the row's intervals).  The binning projects a particle's proxy box onto this plane and
// tests only the packets of the rectangle of columns and rows it reaches, instead of scanning every super tile (1.18 -> ms at 1 M
// particles).  hdr = {a, u, w, ok}: ok = 0 when some ray is 87 degrees or more off the centre direction (no common plane: the super
// tile scan serves the frame).
struct GrtGrid {
    float* hdr;       // [16] a.xyz, u.xyz, w.xyz, ok (uint32 bits)
    float4* rects;    // [blocks] {x0, x1, y0, y1}, widened by rounding margins
    float2* cols;     // [blocks_x] {x0, x1}
    float2* rows;     // [blocks_y] {y0, y1}
};
__host__ __device__ inline GrtGrid grt_block_grid(const GrtCone* block_cones, uint32_t num_blocks, uint32_t gx) {
    GrtGrid g;
    g.hdr = reinterpret_cast<float*>(const_cast<GrtPyramid*>(grt_block_pyramids(block_cones, num_blocks) + num_blocks));
    g.rects = reinterpret_cast<float4*>(g.hdr + 16);
    g.cols = reinterpret_cast<float2*>(g.rects + num_blocks);
    g.rows = g.cols + gx;
    return g;
}
size_t grt_cone_table_bytes(int W, int H);   // cones + pyramids + the frame's tangent-plane tables

// Log of a training forward, so that the backward replays the hits instead of traversing again.  One chunk = what a wave met in one
// trace round, [slot][lane]: the (up to 16) candidates of each ray's round and the round's ghosts — candidates the round was not
// offered because the ray had left their proxy box before the round's tmin — merged in (hit distance, particle) order; ghosts carry
// kGrtGhostBit, unused slots are 0xFFFFFFFF.  A wave's chunks are listed in `table[block][round]`.
// Why ghosts: the reference's backward program (referenceBwdOptix.cu:103-170) traces again with tmax = endT, the last hit distance.
// Hits whose box the ray enters beyond endT are not offered to it (at 1 M particles / 800x800 that happens on 70 % of the rays), its
// rounds of 16 therefore end elsewhere than the forward's, and with other round boundaries other candidates pass the box-exit test
// (tfar >= tmin): processed hits drop out and ghosts come in (8 % of the rays at that size).  Walking a chunk sequence front to back
// with the backward's own intervals (grt_replay_bwd_kernel) reproduces that program exactly: every candidate it can be offered is a
// processed hit or a ghost of the forward.  `ray_flags[ray]` = kGrtRederiveRay where a round saw more ghosts than a chunk holds:
// those rays (and every ray if the pool or the table overflowed: `state[1]`) get their rounds re-derived by grt_trace_bwd_kernel.
constexpr int kGrtMaxGhosts = 8;
constexpr int kGrtLogSlots = kGrtMaxHits + kGrtMaxGhosts;
constexpr uint32_t kGrtGhostBit = 0x80000000u;
constexpr uint32_t kGrtRederiveRay = 0x80000000u;
struct GrtHitLog {
    uint32_t* pool;      // [capacity_chunks][kGrtLogSlots][64]
    uint32_t* table;     // [num_blocks][max_rounds]
    uint32_t* ray_flags; // [W*H]
    uint32_t* state;     // [0] = chunks allocated, [1] = overflow flag
    uint32_t capacity_chunks, max_rounds;
};

// triangle mesh of the hybrid path (playground): its own LBVH over triangle boxes + the per-face / per-vertex attributes, the material
// table (device copy of the caller's GrtMaterial array) and the environment map
struct GrtMeshView {
    const GrtNode* nodes;        // [max(F-1, 1)]
    const float* vertices;       // [V,3]
    const int32_t* triangles;    // [F,3]
    const float* vnormals;       // [V,3]
    const float* vtangents;      // [V,3] or null
    const uint8_t* vhas_tangents;// [V] or null
    const int32_t* prim_type;    // [F] PlaygroundPrimitiveTypes
    const float* mat_uv;         // [F,3,2] or null
    const int32_t* mat_id;       // [F] or null
    const float* refr;           // [F]
    const GrtMaterial* materials;// device
    uint32_t num_materials;
    GrtTexture envmap;
    float envmap_offset[2];
    uint32_t F;
};
struct GrtHybridParams {
    uint32_t opts;               // PlaygroundRenderOptions
    uint32_t max_pbr_bounces;
    uint32_t frame_number;
};

// build stages
void grt_launch_proxies(hipStream_t s, const GrtBuildParams& P, const float* pos, const float* rot, const float* scl, const float* dns,
                        float* inst, float* aabb, float* slack, uint32_t* scene_enc, float* box8);
void grt_launch_morton(hipStream_t s, uint32_t N, const float* aabb, const uint32_t* scene_enc, float* scene, uint32_t* codes, uint32_t* ids);
void grt_launch_hierarchy(hipStream_t s, uint32_t N, const uint32_t* sorted_codes, const uint32_t* sorted_ids, GrtNode* nodes);
void grt_launch_refit(hipStream_t s, uint32_t N, const float* aabb, const float* slack, GrtNode* nodes, uint8_t* done, uint32_t* todo = nullptr);
size_t grt_scene_enc_bytes();
// trace
void grt_launch_trace_fwd(hipStream_t s, const GrtTraceParams& P, const GrtBvh& bvh, const float* density12, const float* sph,
                          const float* ray_o, const float* ray_d, float* out_rad, float* out_dns, float* out_hit2, float* out_nrm,
                          float* out_cnt, int32_t* visibility, uint32_t* dbg_ids, uint32_t* dbg_count, unsigned long long* counters,
                          const GrtHitLog& log, const GrtLists& lists);
// `lists`: the packet lists of the forward this backward belongs to (ranges == nullptr: none — the exact rounds walk the tree);
// the replay goes to stream `s`, the re-derivation of the flagged rays (or of every ray without a log) to `s_rederive`
void grt_launch_trace_bwd(hipStream_t s, hipStream_t s_rederive, const GrtTraceParams& P, const GrtBvh& bvh, const float* density12, const float* sph,
                          const float* ray_o, const float* ray_d, const float* rad, const float* dns, const float* hit2, const float* g_rad,
                          const float* g_dns, const float* g_hit, float* g_density12, float* g_sph, const GrtHitLog& log, const GrtLists& lists);

// neural harmonic features: the ray features of a frame from the forward's hit log (grt_nht_fwd_kernel)
void grt_launch_nht_fwd(hipStream_t s, const GrtTraceParams& P, const float* density12, const float* features, const float* ray_o, const float* ray_d,
                        float* out_feat, const GrtHitLog& log);
// packet lists (GrtLists)
void grt_launch_list_cones(hipStream_t s, const GrtTraceParams& P, const float* ray_o, const float* ray_d, uint32_t* uniform_origin,
                           uint32_t* dir_len_enc /* [2] */, GrtCone* block_cones, GrtCone* super_cones);
void grt_launch_list_count(hipStream_t s, const GrtTraceParams& P, const GrtBvh& bvh, const float* ray_o, const uint32_t* uniform_origin,
                           const uint32_t* dir_len_enc, const GrtCone* block_cones, const GrtCone* super_cones, float* inst_rel, uint32_t* key_bits,
                           uint32_t* counts, uint32_t* particle_idx, void* pair_cache /* grt_pair_cache_bytes(N) */);
void grt_launch_list_expand(hipStream_t s, const GrtTraceParams& P, const GrtBvh& bvh, const float* ray_o, const uint32_t* uniform_origin,
                            const uint32_t* dir_len_enc, const GrtCone* block_cones, const GrtCone* super_cones, const uint32_t* rank_to_particle,
                            const uint32_t* offsets, const uint32_t* counts, uint32_t* starts /* [N] scratch */, uint32_t capacity,
                            uint32_t* block_keys, uint32_t* vals, void* pair_cache);
size_t grt_pair_cache_bytes(uint32_t N);
void grt_launch_list_check(hipStream_t s, uint32_t n, const uint32_t* offsets, uint32_t* flag /* [1] = 1 on overflow */);
void grt_launch_list_ranges(hipStream_t s, uint32_t n, const uint32_t* n_dev /* nullptr: n is the count */, uint32_t num_blocks, const uint32_t* sorted_keys,
                            uint32_t* ranges);
uint32_t grt_num_blocks(int W, int H);
uint32_t grt_num_super(int W, int H);
void grt_launch_mesh_aabb(hipStream_t s, uint32_t F, const float* vertices, const int32_t* triangles, float* aabb, float* slack, uint32_t* scene_enc);
void grt_launch_hybrid(hipStream_t s, const GrtTraceParams& P, const GrtBvh& bvh, const GrtMeshView& mesh, const GrtHybridParams& hp,
                       const float* density12, const float* sph, const float* ray_o, const float* ray_d, const float* ray_max_t, float* out_rgb,
                       float* out_alpha, float* out_last_ray, uint32_t* out_bounces, const GrtLists& lists);

}  // namespace grut

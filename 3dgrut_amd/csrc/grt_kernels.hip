// grt_kernels.hip — 3DGRT device code for gfx950: Gaussian proxy instances, LBVH build, and a software stack-based
// BVH traversal that gathers per-ray ordered hit lists, composites them and differentiates them.
//
// Reference behaviour restated (not translated): threedgrt_tracer/src/particlePrimitives.cu:27-51, 543-610 (proxy
// instances), src/optixTracer.cpp:616-890 (acceleration structure build; OptiX + RT cores there, an LBVH here),
// src/kernels/cuda/referenceOptix.cu:45-248 (k = 16 nearest-hit rounds, any-hit insertion sort),
// referenceBwdOptix.cu:103-170, include/3dgrt/kernels/cuda/gaussianParticles.cuh:337-731 (per-hit math).
//
// CDNA4 design (no RT cores):
//   * LBVH: keys = (size octave, 27-bit Morton code of the proxy centre), the stable radix sort shared with the 3DGUT
//     path (ties broken by particle index), Karras' radix-tree construction, bottom-up refit with agent-scope acquire/release counters.
//     Nodes are 64 B and carry BOTH children's boxes, so each dependent fetch serves two slab tests.
//   * One lane per ray, one wave64 per 8x8 pixel block (coherent rays share node fetches through L1/L2); the per-lane
//     traversal stack lives in LDS ([depth][lane], conflict-free); the 16-entry (distance, particle) buffer of a trace
//     round lives in registers and is kept sorted by an unrolled compare-exchange chain, lexicographic in
//     (distance, particle index) so that the hit order does not depend on traversal order.
//   * A candidate's hit distance t (closest approach in the proxy's scaled frame) can precede the entry of the ray
//     into the proxy's box by at most sqrt(2) * (largest proxy half axis); nodes store that slack, and a subtree is
//     pruned only when (box entry - slack) exceeds the current 16th-nearest distance.
#include <hip/hip_fp16.h>
#include <cstdlib>

#include "grt_internal.hpp"

namespace grut {

namespace {

// ---------------------------------------------------------------------------------------------
// proxies
// ---------------------------------------------------------------------------------------------
// particlePrimitives.cu:27-51 kernelScale
__device__ __forceinline__ float kernel_scale(float density, float min_response, int clamping, float degree) {
    const float modulation = clamping ? density : 1.0f;
    const float mr = fminf(min_response / modulation, 0.97f);
    if (degree < 0.f) {
        const float k = fabsf(degree);
        const float s = 1.0f / powf(3.0f, k);
        return powf((1.f / (logf(mr) - 1.f) + 1.f) / s, 1.f / k);
    }
    if (degree == 0.f) return ((1.0f - mr) / 3.0f) / -0.329630334487f;
    const float a = -4.5f / powf(3.0f, degree);
    return powf(logf(mr) / a, 1.0f / degree);
}

__device__ __forceinline__ uint32_t enc_ordered(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_ordered(uint32_t e) {
    return __uint_as_float((e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e);
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

#include "grt_polyhedra.inl"

constexpr int kSceneReplicas = 64, kSceneReplicaStride = 32;   // one 128-byte line per replica of the encoded scene box

// computeGaussianEnclosingInstancesKernel (particlePrimitives.cu:543-610), emitted as the inverse instance map
__global__ __launch_bounds__(256) void grt_proxy_kernel(GrtBuildParams P, const float* __restrict__ pos, const float* __restrict__ rot,
                                                        const float* __restrict__ scl, const float* __restrict__ dns,
                                                        float* __restrict__ inst, float* __restrict__ aabb, float* __restrict__ slack,
                                                        uint32_t* __restrict__ scene_enc, float* __restrict__ box8) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;   // proxy
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    if (i < P.N) {
        // GRUT_PRIM_TRIHEXA: proxy 3 p + j is the rhombus of particle p in the proxy frame's plane j (x = 0, y = 0, z = 0); its record is
        // the particle's (the candidate test reads the plane from the proxy index)
        // GRUT_PRIM_SPHERE: proxies 2 p and 2 p + 1 are the entry and the exit root of particle p's enclosing sphere (one record for both)
        const bool hexa = P.prim == GRUT_PRIM_TRIHEXA, ball = P.prim == GRUT_PRIM_SPHERE;
        const size_t pi = hexa ? i / 3u : (ball ? i / 2u : i);
        const int plane = hexa ? (int)(i - 3u * (uint32_t)pi) : -1;
        const m3 rt = quat_wxyz_to_rotT(rot[4 * pi], rot[4 * pi + 1], rot[4 * pi + 2], rot[4 * pi + 3]);
        const float ks = kernel_scale(dns[pi], P.min_response, P.clamping, (float)P.degree);
        float k0 = ks * scl[3 * pi], k1 = ks * scl[3 * pi + 1], k2 = ks * scl[3 * pi + 2];
        const float cx = pos[3 * pi], cy = pos[3 * pi + 1], cz = pos[3 * pi + 2];
        float* o = inst + 12 * (size_t)i;
        if (ball) {
            // computeGaussianEnclosingSphereKernel (particlePrimitives.cu:386-403): radius = max(scale) * kernelScale; the record is the map
            // into the frame scaled by it, W = diag(1 / r) (orc_grt_proxies writes the same quotients)
            const float rad = fmaxf(scl[3 * pi], fmaxf(scl[3 * pi + 1], scl[3 * pi + 2])) * ks;
            k0 = k1 = k2 = rad;
            const float ir = 1.f / rad;
            o[0] = ir; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f; o[4] = ir; o[5] = 0.f; o[6] = 0.f; o[7] = 0.f; o[8] = ir;
            o[9] = cx; o[10] = cy; o[11] = cz;
        } else {
        o[0] = rt.r0.x / k0; o[1] = rt.r0.y / k0; o[2] = rt.r0.z / k0;
        o[3] = rt.r1.x / k1; o[4] = rt.r1.y / k1; o[5] = rt.r1.z / k1;
        o[6] = rt.r2.x / k2; o[7] = rt.r2.y / k2; o[8] = rt.r2.z / k2;
        o[9] = cx; o[10] = cy; o[11] = cz;
        }
        // world half extents of the oriented box, padded by a hair so that rounding can never make the ray miss the
        // AABB of a box it touches (culling must stay conservative; candidates are decided in the proxy's own frame)
        // (triangle-mesh proxies: the polyhedron's vertices reach ext_k along axis k of the proxy's frame instead of 1)
        float e0, e1, e2;
        if (hexa) {   // a rhombus (+-sqrt 2 along the two axes of its plane, flat along the third)
            e0 = plane == 0 ? 0.f : 1.4142135381698608f * k0; e1 = plane == 1 ? 0.f : 1.4142135381698608f * k1; e2 = plane == 2 ? 0.f : 1.4142135381698608f * k2;
        } else if (ball) {
            e0 = e1 = e2 = k0;
        } else {
            e0 = k0 * kGrtPolyhedra[P.prim].ext[0]; e1 = k1 * kGrtPolyhedra[P.prim].ext[1]; e2 = k2 * kGrtPolyhedra[P.prim].ext[2];
        }
        float hx = fabsf(rt.r0.x) * e0 + fabsf(rt.r1.x) * e1 + fabsf(rt.r2.x) * e2;
        float hy = fabsf(rt.r0.y) * e0 + fabsf(rt.r1.y) * e1 + fabsf(rt.r2.y) * e2;
        float hz = fabsf(rt.r0.z) * e0 + fabsf(rt.r1.z) * e1 + fabsf(rt.r2.z) * e2;
        if (ball) { hx = k0; hy = k0; hz = k0; }   // (the sphere's own box)
        if (P.prim == GRUT_PRIM_CUSTOM) {
            // computeGaussianEnclosingAABBKernel (particlePrimitives.cu:498-541): min / max over the 8 corners R (c * kscl) + mu, c = +-1 - the
            // extreme corner of an axis has all three products of one sign, so the bound is mu -+ ((|R_c0| k0 + |R_c1| k1) + |R_c2| k2) in the
            // kernel's own (uncontracted, left-to-right) summation order.  THIS box decides which rays the particle is offered to
            // (candidate_abe); the padded copy below only steers the hierarchy.
#pragma clang fp contract(off)
            const float ux = (fabsf(rt.r0.x) * k0 + fabsf(rt.r1.x) * k1) + fabsf(rt.r2.x) * k2;
            const float uy = (fabsf(rt.r0.y) * k0 + fabsf(rt.r1.y) * k1) + fabsf(rt.r2.y) * k2;
            const float uz = (fabsf(rt.r0.z) * k0 + fabsf(rt.r1.z) * k1) + fabsf(rt.r2.z) * k2;
            float* bx = box8 + 8 * (size_t)i;
            bx[0] = cx - ux; bx[1] = cy - uy; bx[2] = cz - uz; bx[3] = cx + ux; bx[4] = cy + uy; bx[5] = cz + uz;
            bx[6] = ks * ks; bx[7] = 0.f;
        }
        hx += 1e-4f * hx + 1e-6f * (fabsf(cx) + 1.f); hy += 1e-4f * hy + 1e-6f * (fabsf(cy) + 1.f); hz += 1e-4f * hz + 1e-6f * (fabsf(cz) + 1.f);
        lo[0] = cx - hx; lo[1] = cy - hy; lo[2] = cz - hz; hi[0] = cx + hx; hi[1] = cy + hy; hi[2] = cz + hz;
        float* b = aabb + 6 * (size_t)i;
        b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = hi[0]; b[4] = hi[1]; b[5] = hi[2];
        // how far the hit distance can precede the ray's entry into the (padded) box: sqrt(2) max kscl for the cube.  Custom primitives: the
        // reported distance t* minimises |W (o + t d - mu)| - the closest approach in the particle's SCALED frame, not in world space - and the
        // ray only has to touch the WORLD box.  With p = o + t_p d any point of the ray inside that box, y(t) = W (p - mu) + (t - t_p) W d
        // and y(t*) perpendicular to W d:  |t* - t_p| <= |W (p - mu)| / |W d| <= kmax sqrt(s0^2 + s1^2 + s2^2),  s_i = sum_j |W_ij| h_j
        // (|p - mu|_j <= h_j, |W d| >= 1 / kmax for a unit direction).  Until round 5 this was the box's half diagonal - the bound of the
        // EUCLIDEAN closest approach, too small by up to the particle's anisotropy: found by the 1 M-particle parity run (one ray in 4 296
        // lost a particle whose t* preceded its box by 1.14 half diagonals; tests/test_grt_custom_slack_cpu.py checks both bounds as mathematics)
        if (P.prim == GRUT_PRIM_CUSTOM) {
            const float s0 = (fabsf(rt.r0.x) * hx + fabsf(rt.r0.y) * hy + fabsf(rt.r0.z) * hz) / k0;
            const float s1 = (fabsf(rt.r1.x) * hx + fabsf(rt.r1.y) * hy + fabsf(rt.r1.z) * hz) / k1;
            const float s2 = (fabsf(rt.r2.x) * hx + fabsf(rt.r2.y) * hy + fabsf(rt.r2.z) * hz) / k2;
            const float kmax = fmaxf(k0, fmaxf(k1, k2));
            // ... and a second bound for the candidates that can be ACCEPTED - the only ones pruning must not lose: the program accepts
            // |W (x* - mu)| ks < 3, i.e. |x* - mu| < 3 kmax / ks = 3 max scl, and p lies within the box's half diagonal of mu:
            // |t* - t_p| = |x* - p| <= 3 max scl + half diagonal.  For needles (kmax / kmin in the hundreds) this is a few kmax where the
            // first bound is hundreds of kmax; the smaller of the two is used.
            const float by_metric = kmax * sqrtf(s0 * s0 + s1 * s1 + s2 * s2);
            const float by_accept = 3.f * (kmax / ks) + sqrtf(hx * hx + hy * hy + hz * hz);
            slack[i] = 1.0001f * fminf(by_metric, by_accept);
        } else {
            slack[i] = 1.41421356237f * 1.0001f * fmaxf(e0, fmaxf(e1, e2));
        }
    }
    // scene box: one set of atomics per wave, spread over kSceneReplicas cache lines (15 k waves hammering six words of
    // ONE line serialised in L2 and cost 1 ms of this kernel's 1.07 ms); the Morton kernel folds the replicas
    uint32_t* rep = scene_enc + (blockIdx.x % kSceneReplicas) * kSceneReplicaStride;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float mn = wave_min(lo[k]), mx = wave_max(hi[k]);
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&rep[k], enc_ordered(mn));
            atomicMax(&rep[3 + k], enc_ordered(mx));
        }
    }
}
__global__ void grt_scene_init_kernel(uint32_t* __restrict__ scene_enc) {
    const uint32_t r = threadIdx.x;
    if (r < (uint32_t)kSceneReplicas)
        for (int k = 0; k < 3; ++k) {
            scene_enc[r * kSceneReplicaStride + k] = 0xFFFFFFFFu;
            scene_enc[r * kSceneReplicaStride + 3 + k] = 0u;
        }
}

__device__ __forceinline__ uint32_t expand_bits9(uint32_t v) {  // 9 bits -> every third bit of 27
    v &= 0x1FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
// Sort key = (size class : 3 bits) | (27-bit Morton code of the proxy centre).  The size class is the octave of the
// proxy's largest half extent relative to the scene: a radix tree over these keys separates the octaves in its top
// three levels, so a few scene-sized Gaussians bloat only the small subtree of their own class instead of the ancestors
// of a million small ones (an LBVH on centres alone made every ray visit thousands of inflated nodes per trace round).
__global__ __launch_bounds__(256) void grt_morton_kernel(uint32_t N, const float* __restrict__ aabb, const uint32_t* __restrict__ scene_enc,
                                                         float* __restrict__ scene, uint32_t* __restrict__ codes, uint32_t* __restrict__ ids) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ float s_scene[6];
    if (threadIdx.x < 64) {  // fold the replicas (kSceneReplicas == 64: one per lane of the first wave)
        const uint32_t* rep = scene_enc + threadIdx.x * kSceneReplicaStride;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float mn = wave_min(dec_ordered(rep[k])), mx = wave_max(dec_ordered(rep[3 + k]));
            if (threadIdx.x == 0) { s_scene[k] = mn; s_scene[3 + k] = mx; }
        }
    }
    __syncthreads();
    float s[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] = s_scene[k];
    if (i == 0)
        for (int k = 0; k < 6; ++k) scene[k] = s[k];
    if (i >= N || !codes) return;  // codes == nullptr: refit-only update, just publish the scene box
    const float* b = aabb + 6 * (size_t)i;
    uint32_t q[3];
    float half_max = 0.f, scene_max = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float c = 0.5f * (b[k] + b[3 + k]);
        const float ext = fmaxf(s[3 + k] - s[k], 1e-30f);
        const float u = fminf(fmaxf((c - s[k]) / ext, 0.f), 1.f);
        q[k] = min((uint32_t)(u * 512.f), 511u);
        half_max = fmaxf(half_max, 0.5f * (b[3 + k] - b[k]));
        scene_max = fmaxf(scene_max, ext);
    }
    const int octave = (int)floorf(log2f(scene_max / fmaxf(half_max, 1e-30f)));  // >= 0: the scene contains the proxy
    const uint32_t cls = (uint32_t)min(max(octave - 1, 0), 7);
    codes[i] = (cls << 27) | (expand_bits9(q[0]) << 2) | (expand_bits9(q[1]) << 1) | expand_bits9(q[2]);
    ids[i] = i;
}

// ---------------------------------------------------------------------------------------------
// hierarchy: Karras 2012 radix tree over the sorted (code, position) keys
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int delta_fn(const uint32_t* __restrict__ codes, int N, int i, int j) {
    if (j < 0 || j >= N) return -1;
    const uint32_t a = codes[i], b = codes[j];
    if (a == b) return 32 + __clz((uint32_t)(i ^ j));
    return __clz(a ^ b);
}
__global__ __launch_bounds__(256) void grt_hierarchy_kernel(uint32_t Nu, const uint32_t* __restrict__ codes, const uint32_t* __restrict__ sorted_ids,
                                                            GrtNode* __restrict__ nodes) {
    const int N = (int)Nu;
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= N - 1) return;
    const int d = (delta_fn(codes, N, i, i + 1) - delta_fn(codes, N, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta_fn(codes, N, i, i - d);
    int lmax = 2;
    while (delta_fn(codes, N, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta_fn(codes, N, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta_fn(codes, N, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (delta_fn(codes, N, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const bool left_leaf = min(i, j) == gamma, right_leaf = max(i, j) == gamma + 1;
    // child codes: internal node index, or leaf bit | particle (the sorted order, hence these codes, survives refit-only updates)
    nodes[i].c[0] = left_leaf ? (kGrtLeafBit | sorted_ids[gamma]) : (uint32_t)gamma;
    nodes[i].c[1] = right_leaf ? (kGrtLeafBit | sorted_ids[gamma + 1]) : (uint32_t)(gamma + 1);
}

__device__ __forceinline__ void write_child(GrtNode* __restrict__ node, uint32_t side, const float lo[3], const float hi[3], float slack) {
    node->lox[side] = lo[0]; node->loy[side] = lo[1]; node->loz[side] = lo[2];
    node->hix[side] = hi[0]; node->hiy[side] = hi[1]; node->hiz[side] = hi[2];
    node->slack[side] = slack;
}
// Bottom-up refit in level-synchronous passes: pass p fills the two child slots of every node whose internal children
// were finished by an EARLIER pass (`done[c]` holds pass + 1; a value written during the current launch is ignored, so
// everything a thread relies on crossed a kernel boundary and is visible on every XCD).  The previous scheme — each leaf
// thread climbing with an agent-scope acquire/release counter per node — spent 4.5 ms at 1 M particles in the cache
// write-backs / invalidates those fences imply; ~height short launches cost a tenth of that.
__device__ __forceinline__ void child_box(const GrtNode* __restrict__ nodes, const float* __restrict__ aabb, const float* __restrict__ slack,
                                          uint32_t c, float lo[3], float hi[3], float& sl) {
    if (c & kGrtLeafBit) {
        const uint32_t id = c & ~kGrtLeafBit;
        const float* b = aabb + 6 * (size_t)id;
        lo[0] = b[0]; lo[1] = b[1]; lo[2] = b[2]; hi[0] = b[3]; hi[1] = b[4]; hi[2] = b[5];
        sl = slack[id];
    } else {
        const float4* q = reinterpret_cast<const float4*>(&nodes[c]);   // {lox, loy | loz, hix | hiy, hiz | c, slack} pairs
        const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        lo[0] = fminf(q0.x, q0.y); lo[1] = fminf(q0.z, q0.w); lo[2] = fminf(q1.x, q1.y);
        hi[0] = fmaxf(q1.z, q1.w); hi[1] = fmaxf(q2.x, q2.y); hi[2] = fmaxf(q2.z, q2.w);
        sl = fmaxf(q3.z, q3.w);
    }
}
// `todo` (last regular pass only): the nodes this pass could not finish either are listed ([0] = count, then the nodes) for
// grt_refit_finish_kernel
__global__ __launch_bounds__(256) void grt_refit_pass_kernel(uint32_t N, uint32_t pass, const float* __restrict__ aabb,
                                                             const float* __restrict__ slack, GrtNode* nodes, uint8_t* done, uint32_t* todo) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float lo[3], hi[3], sl;
    if (N == 1) {  // single particle: the root has one leaf and one empty slot
        if (i == 0 && pass == 0) {
            child_box(nodes, aabb, slack, kGrtLeafBit | 0u, lo, hi, sl);
            write_child(&nodes[0], 0, lo, hi, sl);
            nodes[0].c[0] = kGrtLeafBit | 0u;
            const float elo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, ehi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
            write_child(&nodes[0], 1, elo, ehi, 0.f);
            nodes[0].c[1] = kGrtNoChild;
        }
        return;
    }
    if (i >= N - 1 || done[i] != 0) return;
    const uint32_t c0 = nodes[i].c[0], c1 = nodes[i].c[1];
    const bool r0 = (c0 & kGrtLeafBit) || (done[c0] != 0 && done[c0] <= pass);
    const bool r1 = (c1 & kGrtLeafBit) || (done[c1] != 0 && done[c1] <= pass);
    if (!(r0 && r1)) {
        if (todo) todo[1u + atomicAdd(&todo[0], 1u)] = i;
        return;
    }
    child_box(nodes, aabb, slack, c0, lo, hi, sl);
    write_child(&nodes[i], 0, lo, hi, sl);
    child_box(nodes, aabb, slack, c1, lo, hi, sl);
    write_child(&nodes[i], 1, lo, hi, sl);
    done[i] = (uint8_t)(pass + 1);
}
// The top of the tree in ONE launch.  After kGrtRefitPasses level-synchronous launches only the nodes higher than that are open - a few
// hundred of a million - but the tree may be up to 62 levels deep and every further launch costs ~6 us of dependent launch latency whether it
// finds work or not (60 fixed launches: 0.37 of the 0.51 ms build).  One workgroup finishes the listed nodes pass by pass between workgroup
// barriers.  What an earlier pass of THIS launch wrote is read around the L1 (agent-scope atomic loads: another wave of the CU may hold the
// line of the neighbouring node), and published with a fence before the barrier; the "finished by an EARLIER pass" rule is the launches' own.
constexpr uint32_t kGrtRefitPasses = 20;   // (12: the finisher takes 0.37 ms for the longer list; 20: 0.05 ms)
__device__ __forceinline__ float coherent_load(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void child_box_coherent(const GrtNode* nodes, const float* __restrict__ aabb, const float* __restrict__ slack, uint32_t c,
                                                   float lo[3], float hi[3], float& sl) {
    if (c & kGrtLeafBit) { child_box(nodes, aabb, slack, c, lo, hi, sl); return; }
    const GrtNode* n = &nodes[c];
    lo[0] = fminf(coherent_load(&n->lox[0]), coherent_load(&n->lox[1])); lo[1] = fminf(coherent_load(&n->loy[0]), coherent_load(&n->loy[1]));
    lo[2] = fminf(coherent_load(&n->loz[0]), coherent_load(&n->loz[1]));
    hi[0] = fmaxf(coherent_load(&n->hix[0]), coherent_load(&n->hix[1])); hi[1] = fmaxf(coherent_load(&n->hiy[0]), coherent_load(&n->hiy[1]));
    hi[2] = fmaxf(coherent_load(&n->hiz[0]), coherent_load(&n->hiz[1]));
    sl = fmaxf(coherent_load(&n->slack[0]), coherent_load(&n->slack[1]));
}
__global__ __launch_bounds__(1024) void grt_refit_finish_kernel(uint32_t N, uint32_t first_pass, const float* __restrict__ aabb,
                                                                const float* __restrict__ slack, GrtNode* nodes, uint8_t* done, const uint32_t* todo) {
    const uint32_t n = todo[0];
    // up to kOwn listed nodes per thread live in registers (node, children, open?): a pass is then two dependent round trips to the L2
    // (the children's pass marks, their boxes) and a barrier; longer lists take the same loop from memory
    constexpr int kOwn = 4;
    uint32_t own[kOwn], ch0[kOwn], ch1[kOwn];
    bool is_open[kOwn];
#pragma unroll
    for (int q = 0; q < kOwn; ++q) {
        const uint32_t k = threadIdx.x + (uint32_t)q * blockDim.x;
        is_open[q] = k < n;
        own[q] = is_open[q] ? todo[1u + k] : 0u;
        ch0[q] = is_open[q] ? nodes[own[q]].c[0] : 0u;   // (written by the hierarchy kernel: an earlier launch)
        ch1[q] = is_open[q] ? nodes[own[q]].c[1] : 0u;
    }
    auto try_node = [&](uint32_t i, uint32_t c0, uint32_t c1, uint32_t pass) -> bool {
        const uint8_t d0 = (c0 & kGrtLeafBit) ? 1 : __hip_atomic_load(done + c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint8_t d1 = (c1 & kGrtLeafBit) ? 1 : __hip_atomic_load(done + c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!(d0 != 0 && d0 <= pass && d1 != 0 && d1 <= pass)) return false;
        float lo[3], hi[3], sl;
        child_box_coherent(nodes, aabb, slack, c0, lo, hi, sl);
        write_child(&nodes[i], 0, lo, hi, sl);
        child_box_coherent(nodes, aabb, slack, c1, lo, hi, sl);
        write_child(&nodes[i], 1, lo, hi, sl);
        __hip_atomic_store(done + i, (uint8_t)(pass + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    };
    for (uint32_t pass = first_pass; pass < (uint32_t)kGrtMaxDepth; ++pass) {
        int open = 0;
#pragma unroll
        for (int q = 0; q < kOwn; ++q)
            if (is_open[q]) { is_open[q] = !try_node(own[q], ch0[q], ch1[q], pass); open |= is_open[q] ? 1 : 0; }
        for (uint32_t k = threadIdx.x + (uint32_t)kOwn * blockDim.x; k < n; k += blockDim.x) {
            const uint32_t i = todo[1u + k];
            if (__hip_atomic_load(done + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) continue;
            if (!try_node(i, nodes[i].c[0], nodes[i].c[1], pass)) open = 1;
        }
        __threadfence();
        if (!__syncthreads_or(open)) break;
    }
}

// ---------------------------------------------------------------------------------------------
// per-ray machinery
// ---------------------------------------------------------------------------------------------
struct RayW {
    f3 o, d, inv;
    int prim;   // GrtTraceParams::prim rides with the ray: the candidate test is the one place that depends on it
    int absdist;   // custom primitives under the Slang pipelines (neural harmonic features): the UNSIGNED hit distance of particleDensityHitCustom
    const float* box8;   // GRUT_PRIM_CUSTOM: per particle {world box min, max (as the reference's AABB kernel computes it), kernelScale^2, 0}
};
__device__ __forceinline__ float safe_rcp(float v) {
    return fabsf(v) > 1e-30f ? 1.0f / v : copysignf(1.0e30f, v);
}
// pipelineParameters.h:97-117 (no contraction: the oracle's candidate test starts from the same world-space ray)
__device__ __forceinline__ f3 world_origin(const GrtTraceParams& P, f3 so) {
#pragma clang fp contract(off)
    const float* m = P.ray_to_world_dev ? P.ray_to_world_dev : P.ray_to_world;
    return mk3(m[0] * so.x + m[1] * so.y + m[2] * so.z + m[3], m[4] * so.x + m[5] * so.y + m[6] * so.z + m[7],
               m[8] * so.x + m[9] * so.y + m[10] * so.z + m[11]);
}
__device__ __forceinline__ RayW make_ray(const GrtTraceParams& P, const float* __restrict__ ray_o, const float* __restrict__ ray_d, size_t pix) {
    const f3 so = mk3(ray_o[3 * pix], ray_o[3 * pix + 1], ray_o[3 * pix + 2]);
    const f3 sd = mk3(ray_d[3 * pix], ray_d[3 * pix + 1], ray_d[3 * pix + 2]);
    const float* m = P.ray_to_world_dev ? P.ray_to_world_dev : P.ray_to_world;
    RayW r;
    r.o = world_origin(P, so);
    {
#pragma clang fp contract(off)
        r.d = mk3(m[0] * sd.x + m[1] * sd.y + m[2] * sd.z, m[4] * sd.x + m[5] * sd.y + m[6] * sd.z, m[8] * sd.x + m[9] * sd.y + m[10] * sd.z);
    }
    r.inv = mk3(safe_rcp(r.d.x), safe_rcp(r.d.y), safe_rcp(r.d.z));
    r.prim = P.prim;
    r.absdist = (P.prim == GRUT_PRIM_CUSTOM && P.nht) ? 1 : 0;
    r.box8 = P.box8;
    return r;
}
// referenceOptix.cu:33-39 intersectAABB
__device__ __forceinline__ void scene_interval(const float* __restrict__ s, const RayW& r, float& tmin, float& tmax) {
#pragma clang fp contract(off)
    const float t0x = (s[0] - r.o.x) / r.d.x, t0y = (s[1] - r.o.y) / r.d.y, t0z = (s[2] - r.o.z) / r.d.z;
    const float t1x = (s[3] - r.o.x) / r.d.x, t1y = (s[4] - r.o.y) / r.d.y, t1z = (s[5] - r.o.z) / r.d.z;
    tmin = fmaxf(0.f, fmaxf(fminf(t0x, t1x), fmaxf(fminf(t0y, t1y), fminf(t0z, t1z))));
    tmax = fminf(fmaxf(t0x, t1x), fminf(fmaxf(t0y, t1y), fmaxf(t0z, t1z)));
}

// candidate test in the proxy's frame — written with explicit operation order (fused multiply-adds spelled out, no contraction
// of anything else), so that the CPU checker (tests) can evaluate bit-identical distances and the per-ray hit ORDER can be compared
// exactly.  Round 3 rewrote the sequence for the instruction count (the test is a third of the forward): fused dot products, one
// division for the distance, one reciprocal per axis instead of two divisions, v_min / v_max (IEEE minNum / maxNum), and the
// 3-sigma test without the normalisation — 185 -> ~90 instructions for a full test; the checker was rewritten with it, operation
// by operation.
struct Cand {
    float t, tnear, tfar;
    bool ok;
    bool box;   // `always_box` calls: the ray meets the proxy (whatever its hit distance)
    int why;   // instrumented builds: 1 = distance outside the wanted range, 2 = the ray misses the proxy's box, 3 = farther than 3 sigma
};
// `t_lo` / (`t_hi`, `id_hi`): the caller only wants candidates with t_lo < t and (t, id) < (t_hi, id_hi); the hit distance
// is evaluated first and the (division-heavy) box test only for those — the values themselves are unaffected.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef const f4v __attribute__((address_space(4))) cfloat4;   // constant address space: uniform addresses load through the scalar cache
__device__ __forceinline__ float4 ld4(const float4* p, int k) { return p[k]; }
__device__ __forceinline__ float4 ld4(const cfloat4* p, int k) { const f4v v = p[k]; return make_float4(v.x, v.y, v.z, v.w); }
// Q = const float4 (per-lane record) or cfloat4 (wave-uniform record of a read-only array: fetched once per wave)
// REL: the record is a frame-relative one {W rows, W (o - mu)} (GrtBvh::inst_rel) — the proxy-frame origin is read, not computed
template <typename Q, bool REL = false, bool TIES = false>
__device__ __forceinline__ Cand candidate_q(const Q* __restrict__ rec, const RayW& r, float t_lo, float t_hi, uint32_t id, uint32_t id_hi);
__device__ __forceinline__ Cand candidate(const float* __restrict__ inst, const RayW& r, float t_lo = -3.0e38f, float t_hi = 3.0e38f,
                                          uint32_t id = 0u, uint32_t id_hi = 0xFFFFFFFFu) {
    return candidate_q(reinterpret_cast<const float4*>(inst), r, t_lo, t_hi, id, id_hi);
}
template <bool TIES = false>
__device__ __forceinline__ Cand candidate_uniform(const float* inst, const RayW& r, float t_lo, float t_hi, uint32_t id, uint32_t id_hi) {
    return candidate_q<cfloat4, false, TIES>(reinterpret_cast<const cfloat4*>(reinterpret_cast<uintptr_t>(inst)), r, t_lo, t_hi, id, id_hi);
}
// proxy-frame origin W (o - mu) of one record: the ONE definition shared by the candidate test and the frame-relative tables
__device__ __forceinline__ f3 proxy_origin(const float4& a, const float4& b, const float4& e, f3 o) {
#pragma clang fp contract(off)
    const float dlx = o.x - e.y, dly = o.y - e.z, dlz = o.z - e.w;
    return mk3(a.x * dlx + a.y * dly + a.z * dlz, a.w * dlx + b.x * dly + b.y * dlz, b.z * dlx + b.w * dly + e.x * dlz);
}
// (TIES: candidates AT t_lo are evaluated too — t_lo is then the ray's last hit distance, see GhostLog)
template <bool REL, bool TIES = false>
__device__ __forceinline__ Cand candidate_abe(const float4& a, const float4& b, const float4& e, const RayW& r, float t_lo, float t_hi, uint32_t id, uint32_t id_hi,
                                              bool always_box = false, bool may_skip = false);
template <typename Q, bool REL, bool TIES>
__device__ __forceinline__ Cand candidate_q(const Q* __restrict__ rec, const RayW& r, float t_lo, float t_hi, uint32_t id, uint32_t id_hi) {
    const float4 a = ld4(rec, 0), b = ld4(rec, 1), e = ld4(rec, 2);
    return candidate_abe<REL, TIES>(a, b, e, r, t_lo, t_hi, id, id_hi);
}
__device__ __forceinline__ float max3f(float a, float b, float c);
__device__ __forceinline__ float min3f(float a, float b, float c);
// always_box (wave-uniform): the box test runs for rays outside the wanted range too and its outcome is reported in `box`; `ok` and
// everything else are what the plain call returns
// Clip a proxy-frame ray against the face planes n . x <= h of polyhedron PRIM, fully unrolled (the coefficients are instruction literals):
// entering planes (n . d < 0) raise tin, leaving planes lower tout, parallel-and-outside = miss.  Antipodal pairs (n, h) / (-n, h) share their
// two dot products and ONE correctly rounded reciprocal (1 / (-dn) = -(1 / dn) exactly).  The selects pick exactly what the CPU checker's
// plane-by-plane branches assign.  Returns true when `may_skip` let it stop early: the distances only tighten from plane to plane, so a
// ray that has missed (tin > tout), whose entry is beyond the wanted range (tin > t_hi) or whose exit is before it (tout < t_lo) cannot
// become a candidate - when that holds for every ray of the wave the remaining planes are skipped (first tests, always_box, want the true
// `box`: then only the miss counts).
template <int PRIM>
__device__ __forceinline__ bool clip_polyhedron(float pox, float poy, float poz, float pdx, float pdy, float pdz, float t_lo, float t_hi, bool always_box,
                                                bool may_skip, float& tin, float& tout, bool& miss) {
#pragma clang fp contract(off)
    constexpr int kPairs = kGrtPolyhedraCx[PRIM].num_pairs, kPlanes = kGrtPolyhedraCx[PRIM].num_planes;
#pragma unroll
    for (int p = 0; p < kPairs; ++p) {
        const float nx = kGrtPolyhedraCx[PRIM].planes[2 * p][0], ny = kGrtPolyhedraCx[PRIM].planes[2 * p][1], nz = kGrtPolyhedraCx[PRIM].planes[2 * p][2],
                    hh = kGrtPolyhedraCx[PRIM].planes[2 * p][3];
        const float dn = fmaf(nz, pdz, fmaf(ny, pdy, nx * pdx));
        const float dt = fmaf(nz, poz, fmaf(ny, poy, nx * pox));
        const float on_a = hh - dt, on_b = hh + dt;
        const float inv = 1.f / dn;
        const float tf_a = on_a * inv, tf_b = -(on_b * inv);
        const bool neg = dn < 0.f, cut = neg || (dn > 0.f);
        const float t_en = neg ? tf_a : tf_b, t_le = neg ? tf_b : tf_a;
        tin = cut ? fmaxf(tin, t_en) : tin;
        tout = cut ? fminf(tout, t_le) : tout;
        miss = miss || (!cut && (on_a < 0.f || on_b < 0.f));
        if (may_skip && (p % 3 == 2) && (2 * p + 2 < kPlanes) && __all(miss || !(tin <= tout) || (!always_box && ((tin > t_hi) || (tout < t_lo))))) return true;
    }
#pragma unroll
    for (int f = 2 * kPairs; f < kPlanes; ++f) {
        const float nx = kGrtPolyhedraCx[PRIM].planes[f][0], ny = kGrtPolyhedraCx[PRIM].planes[f][1], nz = kGrtPolyhedraCx[PRIM].planes[f][2], hh = kGrtPolyhedraCx[PRIM].planes[f][3];
        const float dn = fmaf(nz, pdz, fmaf(ny, pdy, nx * pdx));
        const float on = hh - fmaf(nz, poz, fmaf(ny, poy, nx * pox));
        const float tf = on * (1.f / dn);
        tin = dn < 0.f ? fmaxf(tin, tf) : tin;
        tout = dn > 0.f ? fminf(tout, tf) : tout;
        miss = miss || (!(dn < 0.f) && !(dn > 0.f) && on < 0.f);
    }
    return false;
}
// may_skip (the rounds' scans only - callers that read `t` of a candidate that is not `ok` must leave it off): the mesh proxies' plane loop may
// end early when no ray of the wave can become a candidate any more (see there)
template <bool REL, bool TIES>
__device__ __forceinline__ Cand candidate_abe(const float4& a, const float4& b, const float4& e, const RayW& r, float t_lo, float t_hi, uint32_t id, uint32_t id_hi,
                                              bool always_box, bool may_skip) {
#pragma clang fp contract(off)
    Cand c;
    c.ok = false; c.box = false; c.t = 0.f; c.tnear = 0.f; c.tfar = 0.f; c.why = 1;
    // inst = {W00 W01 W02 W10 | W11 W12 W20 W21 | W22 mux muy muz}
    const f3 po = REL ? mk3(e.y, e.z, e.w) : proxy_origin(a, b, e, r.o);
    const float pox = po.x, poy = po.y, poz = po.z;
    const float pdx = fmaf(a.z, r.d.z, fmaf(a.y, r.d.y, a.x * r.d.x)), pdy = fmaf(b.y, r.d.z, fmaf(b.x, r.d.y, a.w * r.d.x)),
                pdz = fmaf(e.x, r.d.z, fmaf(b.w, r.d.y, b.z * r.d.x));
    if (r.prim >= GRUT_PRIM_ICOSAHEDRON && r.prim <= GRUT_PRIM_DIAMOND) {
        // triangle-mesh proxies (icosahedron ...): the distance at which the ray ENTERS the fixed convex polyhedron of the proxy's frame -
        // what OptiX reports for the one front-facing triangle of the reference's mesh the ray passes (back faces culled; a ray that starts
        // inside is not offered the particle).  Clip against the face planes n . x <= h: entering planes (n . d < 0) raise the entry
        // distance, leaving planes lower the exit distance; parallel and outside = miss.  Same operations, same order in the CPU checker.
        float tin = -3.0e38f, tout = 3.0e38f;
        bool miss = false;
        // (round 6) the plane loop unrolled per primitive over the constexpr copy of the table (clip_polyhedron): the rolled loop that used to
        // stand here fetched every pair through the scalar cache and waited for it (s_load_dwordx4 + s_waitcnt per pair) and branched on the
        // sign of n . d - the icosahedra's forward went from 11.9 to 8.0 ms
        bool skipped = false;
        switch (r.prim) {
        case GRUT_PRIM_ICOSAHEDRON: skipped = clip_polyhedron<GRUT_PRIM_ICOSAHEDRON>(pox, poy, poz, pdx, pdy, pdz, t_lo, t_hi, always_box, may_skip, tin, tout, miss); break;
        case GRUT_PRIM_OCTAHEDRON: skipped = clip_polyhedron<GRUT_PRIM_OCTAHEDRON>(pox, poy, poz, pdx, pdy, pdz, t_lo, t_hi, always_box, may_skip, tin, tout, miss); break;
        case GRUT_PRIM_TETRAHEDRON: skipped = clip_polyhedron<GRUT_PRIM_TETRAHEDRON>(pox, poy, poz, pdx, pdy, pdz, t_lo, t_hi, always_box, may_skip, tin, tout, miss); break;
        default: skipped = clip_polyhedron<GRUT_PRIM_DIAMOND>(pox, poy, poz, pdx, pdy, pdz, t_lo, t_hi, always_box, may_skip, tin, tout, miss); break;
        }
        if (skipped) { c.t = tin; c.tnear = tin; c.tfar = 3.0e38f; return c; }
        c.t = tin;
        c.box = !miss && (tin <= tout) && (tin > -3.0e38f);
        const bool want = (TIES ? (c.t >= t_lo) : (c.t > t_lo)) && ((c.t < t_hi) || (c.t == t_hi && id < id_hi));
        c.tnear = tin; c.tfar = 3.0e38f;   // the reported distance IS the geometric hit: eligibility is t in (tmin, tmax) alone
        c.why = c.box ? 1 : 2;
        c.ok = c.box && want;
        return c;
    }
    if (r.prim == GRUT_PRIM_TRIHEXA) {
        // trihexa (particlePrimitives.cu:107-153): proxy `id` is the rhombus |a| + |b| <= sqrt 2 of particle id / 3 in the proxy frame's plane
        // id % 3, traced as triangles with back faces culled: the windings make the x = 0 rhombus face +x, the y = 0 rhombus +y, and the two
        // triangles of the z = 0 rhombus face opposite ways (its x >= 0 half +z, its x <= 0 half -z).  Same operations, same order as the CPU
        // checker's trihexa_candidates.
        const uint32_t plane = id % 3u;
        const float pn = plane == 0u ? pdx : (plane == 1u ? pdy : pdz), on = plane == 0u ? pox : (plane == 1u ? poy : poz);
        const bool facing = plane == 2u ? (pn != 0.f) : (pn < 0.f);
        if (!facing) { c.why = 2; return c; }
        const float t = -on / pn;
        // the two in-plane coordinates of the crossing: (y, z) / (x, z) / (x, y)
        const float a0 = plane == 0u ? fmaf(t, pdy, poy) : fmaf(t, pdx, pox);
        const float a1 = plane == 2u ? fmaf(t, pdy, poy) : fmaf(t, pdz, poz);
        c.t = t; c.tnear = t; c.tfar = 3.0e38f;
        c.box = fabsf(a0) + fabsf(a1) <= 1.4142135381698608f;
        if (plane == 2u) c.box = c.box && ((a0 > 0.f && pdz < 0.f) || (a0 < 0.f && pdz > 0.f));
        const bool want = (TIES ? (c.t >= t_lo) : (c.t > t_lo)) && ((c.t < t_hi) || (c.t == t_hi && id < id_hi));
        c.why = c.box ? 1 : 2;
        c.ok = c.box && want;
        return c;
    }
    if (r.prim == GRUT_PRIM_SPHERE) {
        // sphere (optixTracer.cpp:765-781, 823-833): proxy `id` is the entry (even) or the exit (odd) of the ray through the unit sphere of the
        // frame scaled by the particle's radius - the two offers OptiX's built-in sphere intersector makes to an any-hit program that ignores
        // them (include/grut_amd.h).  Same operations, same order as the CPU checker's sphere_candidates and the emulated OptiX.
        const float qa = fmaf(pdz, pdz, fmaf(pdy, pdy, pdx * pdx)), qb = fmaf(poz, pdz, fmaf(poy, pdy, pox * pdx));
        const float qc = fmaf(poz, poz, fmaf(poy, poy, pox * pox)) - 1.f;
        const float disc = fmaf(qb, qb, -(qa * qc));
        if (!(disc >= 0.f) || !(qa > 0.f)) { c.why = 2; return c; }
        const float sq = sqrtf(disc);
        const float t = (id & 1u) ? (-qb + sq) / qa : (-qb - sq) / qa;
        c.t = t; c.tnear = t; c.tfar = 3.0e38f;
        c.box = true;
        const bool want = (TIES ? (c.t >= t_lo) : (c.t > t_lo)) && ((c.t < t_hi) || (c.t == t_hi && id < id_hi));
        c.why = 1;
        c.ok = want;
        return c;
    }
    if (r.prim == GRUT_PRIM_TRISURFEL) {
        // trisurfel (particlePrimitives.cu:155-205): two triangles = the rhombus |x| + |y| <= sqrt 2 of the proxy's z = 0 plane, traced WITHOUT
        // face culling (referenceOptix.cu:62: SurfelPrimitive -> OPTIX_RAY_FLAG_NONE): the reported distance is the plane crossing's.  Same
        // operations, same order in the CPU checker (its grt_oracle.c: candidate, g_prim 6).
        if (pdz == 0.f) { c.why = 2; return c; }
        const float t = -poz / pdz;
        const float hx = fmaf(t, pdx, pox), hy = fmaf(t, pdy, poy);
        c.t = t; c.tnear = t; c.tfar = 3.0e38f;
        c.box = fabsf(hx) + fabsf(hy) <= 1.4142135381698608f;
        const bool want = (TIES ? (c.t >= t_lo) : (c.t > t_lo)) && ((c.t < t_hi) || (c.t == t_hi && id < id_hi));
        c.why = c.box ? 1 : 2;
        c.ok = c.box && want;
        return c;
    }
    // intersectInstanceParticle: hit distance = closest approach in the proxy's frame
    const float numerator = -fmaf(poz, pdz, fmaf(poy, pdy, pox * pdx));
    const float dd = fmaf(pdz, pdz, fmaf(pdy, pdy, pdx * pdx));
    c.t = numerator / dd;
    // (custom primitives with neural harmonic features: the Slang pipeline's intersection test, particleDensityHitCustom,
    // gaussianParticles.slang:489-523, reports canonicalRayDistance - a length - where intersectCustomParticle reports it with the ray
    // parameter's sign: a particle whose maximum lies behind the ray origin is a candidate there.  Folds away for every other primitive.)
    if (r.prim == GRUT_PRIM_CUSTOM && r.absdist) c.t = fabsf(c.t);
    const bool wanted = (TIES ? (c.t >= t_lo) : (c.t > t_lo)) && ((c.t < t_hi) || (c.t == t_hi && id < id_hi));
    if (!wanted && !always_box) return c;
    // (round 6: in the packet lists' test too - the default configuration's forward is an instantiation of its own in which r.prim is a
    // constant and this branch folds away; until then custom primitives walked the tree because the branch cost the instances' forward 3 %)
    if (r.prim == GRUT_PRIM_CUSTOM) {
        // custom primitives (render.primitive_type custom; optixTracer.cpp:638-655, intersectCustomParticle gaussianParticles.cuh:407-441): the
        // intersection program runs for rays that overlap the particle's WORLD box, reports the point of maximum response - the same point
        // as the instances' (the proxy frame differs from the program's scale frame by the scalar kernelScale, which cancels in the
        // distance) - and accepts it within 3 sigma of the SCALE frame: |pd x po|^2 ks^2 < 9 |pd|^2 in proxy-frame quantities.
        // (REL - the packet lists' test: `id` is wave-uniform and the boxes are read-only during a trace, so the record comes through the
        // scalar cache into SGPRs, requested before the arithmetic above needs it; seven vector loads of one address and their wait
        // per test otherwise)
        float bx[8];
        if (REL) {
            const cfloat4* bq = reinterpret_cast<const cfloat4*>(reinterpret_cast<uintptr_t>(r.box8 + 8 * (size_t)__builtin_amdgcn_readfirstlane((int)id)));
            const float4 b0 = ld4(bq, 0), b1 = ld4(bq, 1);
            bx[0] = b0.x; bx[1] = b0.y; bx[2] = b0.z; bx[3] = b0.w; bx[4] = b1.x; bx[5] = b1.y; bx[6] = b1.z; bx[7] = b1.w;
        } else {
            const float4* bq = reinterpret_cast<const float4*>(r.box8 + 8 * (size_t)id);
            const float4 b0 = bq[0], b1 = bq[1];
            bx[0] = b0.x; bx[1] = b0.y; bx[2] = b0.z; bx[3] = b0.w; bx[4] = b1.x; bx[5] = b1.y; bx[6] = b1.z; bx[7] = b1.w;
        }
        const float ax0 = (bx[0] - r.o.x) * r.inv.x, ax1 = (bx[3] - r.o.x) * r.inv.x, ay0 = (bx[1] - r.o.y) * r.inv.y, ay1 = (bx[4] - r.o.y) * r.inv.y;
        const float az0 = (bx[2] - r.o.z) * r.inv.z, az1 = (bx[5] - r.o.z) * r.inv.z;
        const float tnear = max3f(fminf(ax0, ax1), fminf(ay0, ay1), fminf(az0, az1));
        const float tfar = min3f(fmaxf(ax0, ax1), fmaxf(ay0, ay1), fmaxf(az0, az1));
        if (wanted) c.why = 2;
        if (!(tnear <= tfar)) return c;
        if (wanted) { c.why = 3; c.tnear = tnear; c.tfar = tfar; }
        const float crx = fmaf(pdy, poz, -(pdz * poy)), cry = fmaf(pdz, pox, -(pdx * poz)), crz = fmaf(pdx, poy, -(pdy * pox));
        c.box = fmaf(crz, crz, fmaf(cry, cry, crx * crx)) * bx[6] < 9.0f * dd;
        c.ok = c.box && wanted;
        return c;
    }
    // slab test of the unit box: the three reciprocals from ONE correctly rounded division - 1 / (pdx pdy pdz) times the product of the other
    // two components (round 6: 20 instead of 39 instructions; the box test is a fifth of the forward's issue time) - unless that product
    // leaves [1e-24, 1e24] (a ray within rounding of perpendicular to a proxy axis, a zero component): then one division per axis and the
    // planes as (plane - po) * reciprocal, as before (an infinite reciprocal must meet a finite factor).
    // Every intermediate of the short form is a normal number.  IEEE minNum / maxNum.  Same operations, same order in the CPU checker.
    float ax0, ax1, ay0, ay1, az0, az1;
    {
        const float pxy = pdx * pdy, prod = pxy * pdz, aprod = fabsf(prod);
        if (aprod >= 1e-24f && aprod <= 1e24f) {
            const float q = 1.f / prod;
            const float ix = q * (pdy * pdz), iy = q * (pdx * pdz), iz = q * pxy;
            // the slab planes (-1 - po) i and (1 - po) i as one fused multiply-add each (finite reciprocals here)
            ax0 = fmaf(-pox, ix, -ix); ax1 = fmaf(-pox, ix, ix); ay0 = fmaf(-poy, iy, -iy); ay1 = fmaf(-poy, iy, iy); az0 = fmaf(-poz, iz, -iz); az1 = fmaf(-poz, iz, iz);
        } else {
            const float ix = 1.f / pdx, iy = 1.f / pdy, iz = 1.f / pdz;
            ax0 = (-1.f - pox) * ix; ax1 = (1.f - pox) * ix; ay0 = (-1.f - poy) * iy; ay1 = (1.f - poy) * iy; az0 = (-1.f - poz) * iz; az1 = (1.f - poz) * iz;
        }
    }
    const float tnear = max3f(fminf(ax0, ax1), fminf(ay0, ay1), fminf(az0, az1));
    const float tfar = min3f(fmaxf(ax0, ax1), fmaxf(ay0, ay1), fmaxf(az0, az1));
    if (wanted) c.why = 2;
    if (!(tnear <= tfar)) return c;
    if (wanted) { c.why = 3; c.tnear = tnear; c.tfar = tfar; }
    // hitMaxParticleSquaredDistance (pipelineParameters.h:71): |normalize(pd) x po|^2 / |pd|^2 < 9  <=>  |pd x po|^2 < 9 |pd|^4
    const float crx = fmaf(pdy, poz, -(pdz * poy)), cry = fmaf(pdz, pox, -(pdx * poz)), crz = fmaf(pdx, poy, -(pdy * pox));
    c.box = fmaf(crz, crz, fmaf(cry, cry, crx * crx)) < 9.0f * (dd * dd);
    c.ok = c.box && wanted;
    return c;
}

__device__ __forceinline__ bool hit_less(float t, uint32_t id, float bt, uint32_t bi) { return (t < bt) || (t == bt && id < bi); }

// G nearest candidates, ascending.  G = 16 is one trace of the reference; the forward gathers G = 32 per traversal and
// carves two 16-hit rounds out of it (see grt_trace_fwd_kernel).
// The per-ray buffer of one trace: the G nearest candidates in (distance, particle) order — the payload registers and the
// compare-exchange chain of __anyhit__ah (referenceOptix.cu:210-246).  A slot is ONE ordered word: the distance widened to a double
// (exact; a float leaves the low 29 mantissa bits of the double zero) with the particle index in those 29 bits — for the positive
// distances a candidate has, doubles compare like the pair (distance, particle) does lexicographically.  Inserting K into the sorted
// slots a[0..G) is then a[k] = max(a[k-1], min(a[k], K)): two instructions per slot instead of the five compares and four selects of
// the pairwise chain (144 -> 32 per insertion; the forward makes 5.4 M wave-level insertions per frame at 1 M particles).
constexpr uint32_t kGrtHitIdMask = 0x1FFFFFFFu;   // (grt_validate refuses more than 2^29 - 2 particles)
__device__ __forceinline__ double hit_min(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double hit_max(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <int G>
struct HitBufferT {
    struct Last { float v; __device__ __forceinline__ float operator[](int) const { return v; } };      // buf.t[G - 1]
    struct LastId { uint32_t v; __device__ __forceinline__ uint32_t operator[](int) const { return v; } };
    double key[G];
    Last t;        // distance of the farthest slot (3.0e38f while the buffer is not full): the only one the rounds read
    LastId id;     // ... and its particle (0xFFFFFFFF: empty)
    static __device__ __forceinline__ double make(float ht, uint32_t hid) {
        return __longlong_as_double(__double_as_longlong((double)ht) | (long long)(hid & kGrtHitIdMask));
    }
    static __device__ __forceinline__ float key_t(double k) { return (float)__longlong_as_double(__double_as_longlong(k) & ~(long long)kGrtHitIdMask); }
    static __device__ __forceinline__ uint32_t key_id(double k) {
        const uint32_t v = (uint32_t)__double_as_longlong(k) & kGrtHitIdMask;
        return v == kGrtHitIdMask ? 0xFFFFFFFFu : v;
    }
    __device__ __forceinline__ void clear() {
        const double empty = make(3.0e38f, kGrtHitIdMask);
#pragma unroll
        for (int k = 0; k < G; ++k) key[k] = empty;
        t.v = 3.0e38f; id.v = 0xFFFFFFFFu;
    }
    __device__ __forceinline__ uint32_t first_id() const { return key_id(key[0]); }
    // park the sorted list in LDS ([slot][lane]) so that the per-hit code can loop over it instead of being unrolled
    __device__ __forceinline__ void store(float* __restrict__ st, uint32_t* __restrict__ sid, int lane) const {
#pragma unroll
        for (int k = 0; k < G; ++k) { st[k * 64 + lane] = key_t(key[k]); sid[k * 64 + lane] = key_id(key[k]); }
    }
    __device__ __forceinline__ void insert(float ht, uint32_t hid) {
        const double K = make(ht, hid);
#pragma unroll
        for (int k = G - 1; k >= 1; --k) key[k] = hit_max(key[k - 1], hit_min(key[k], K));
        key[0] = hit_min(key[0], K);
        t.v = key_t(key[G - 1]); id.v = key_id(key[G - 1]);
    }
};
using HitBuffer = HitBufferT<kGrtMaxHits>;

// slab test of BOTH children of a node (GrtNode: q0 = {lo0.x, lo1.x, lo0.y, lo1.y}, q1 = {lo0.z, lo1.z, hi0.x, hi1.x},
// q2 = {hi0.y, hi1.y, hi0.z, hi1.z}) in packed fp32: (plane - origin) * inverse direction, one instruction per plane pair
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float min3f(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ void boxes_hit(const float4& q0, const float4& q1, const float4& q2, const RayW& r, float& tn0, float& tf0, float& tn1,
                                          float& tf1, bool& ok0, bool& ok1) {
    const v2f lx = {q0.x, q0.y}, ly = {q0.z, q0.w}, lz = {q1.x, q1.y}, hx = {q1.z, q1.w}, hy = {q2.x, q2.y}, hz = {q2.z, q2.w};
    const v2f ox = splat(r.o.x), oy = splat(r.o.y), oz = splat(r.o.z), ix = splat(r.inv.x), iy = splat(r.inv.y), iz = splat(r.inv.z);
    const v2f x0 = (lx - ox) * ix, x1 = (hx - ox) * ix, y0 = (ly - oy) * iy, y1 = (hy - oy) * iy, z0 = (lz - oz) * iz, z1 = (hz - oz) * iz;
    tn0 = max3f(fminf(x0.x, x1.x), fminf(y0.x, y1.x), fminf(z0.x, z1.x));
    tf0 = min3f(fmaxf(x0.x, x1.x), fmaxf(y0.x, y1.x), fmaxf(z0.x, z1.x));
    tn1 = max3f(fminf(x0.y, x1.y), fminf(y0.y, y1.y), fminf(z0.y, z1.y));
    tf1 = min3f(fmaxf(x0.y, x1.y), fmaxf(y0.y, y1.y), fmaxf(z0.y, z1.y));
    ok0 = tn0 <= tf0 * 1.0000004f + 1e-30f;
    ok1 = tn1 <= tf1 * 1.0000004f + 1e-30f;
}

// 8x8 pixel block of this workgroup.  The image is cut into super-tiles of 8x8 pixel blocks (64x64 pixels); workgroups
// are dealt to the 8 XCDs round-robin by linear id, so XCD k is given super-tiles k, k+8, ... whole: the blocks resident on
// one XCD are 2-D neighbours that share BVH nodes and particles in that XCD's L2, while every XCD still gets super-tiles
// from all over the image (a contiguous band per XCD leaves the XCDs that got empty sky idle).
#ifndef GRT_SUPER_TILE
#define GRT_SUPER_TILE 8
#endif
constexpr int kSuperTile = GRT_SUPER_TILE;  // pixel blocks per side
struct PixelBlock {
    int bx, by;
    uint32_t index;   // row-major block index (hit log key)
    bool inside;
};
__host__ __device__ __forceinline__ uint32_t pixel_block_grid(int W, int H) {
    const uint32_t sx = (uint32_t)(W + 8 * kSuperTile - 1) / (8 * kSuperTile), sy = (uint32_t)(H + 8 * kSuperTile - 1) / (8 * kSuperTile);
    const uint32_t st = (sx * sy + 7u) & ~7u;   // whole rounds of 8 super-tiles
    return st * kSuperTile * kSuperTile;
}
__device__ __forceinline__ PixelBlock pixel_block(int W, int H) {
    const uint32_t gx = (uint32_t)(W + 7) / 8, gy = (uint32_t)(H + 7) / 8;
    const uint32_t sx = (gx + kSuperTile - 1) / kSuperTile;
    const uint32_t b = blockIdx.x, xcd = b & 7u, idx = b >> 3;
    const uint32_t st = (idx / (kSuperTile * kSuperTile)) * 8u + xcd, within = idx % (kSuperTile * kSuperTile);   // (a strided super-tile order: +5 %)
    PixelBlock pb;
    pb.bx = (int)((st % sx) * kSuperTile + (within % kSuperTile));
    pb.by = (int)((st / sx) * kSuperTile + (within / kSuperTile));
    pb.inside = (uint32_t)pb.bx < gx && (uint32_t)pb.by < gy;
    pb.index = (uint32_t)pb.by * gx + (uint32_t)pb.bx;
    return pb;
}

// "Ghosts" of a trace round (training forwards only): candidates the round was NOT offered because the ray had left their proxy box
// before the round's tmin (tfar < tmin) although their hit distance lies in the round's range.  The reference's backward program
// traces with other round boundaries (its rounds skip the hits whose box the ray enters beyond endT, referenceBwdOptix.cu:123-131), so a
// ghost of the forward can be a hit of the backward: the forward logs them next to the processed hits (GrtHitLog) and the replay decides
// with the backward's own intervals.  Per lane: up to kGrtMaxGhosts per round, parked in LDS ([slot][lane]); `n` counts all that were
// seen (n > kGrtMaxGhosts: the ray is flagged and its backward rounds are re-derived instead).
// Only ghosts the backward can possibly be offered are kept: its trace that reaches a ghost of forward round q starts no earlier than
// forward round q - 2 did unless more than 16 hits of rounds q - 2 and q - 1 are missing from its own sequence, so a ghost whose box the
// ray left before THAT round's tmin (`min_tfar`) stays a ghost (the replay checks the premise per ray and counts the rays it fails for:
// GrtHitLog::state[2], expected 0).
// A second kind of ghost: a candidate whose hit distance EQUALS that of the 16th hit of a full round (larger particle index) is the 17th
// of that round and fails t > tmin in the next one — skipped for good by the forward, but not necessarily by a backward whose rounds
// end elsewhere.  The round after (tie_t, tie_id) = the last processed hit logs those as its first entries.
struct GhostLog {
    uint32_t* id;    // LDS [kGrtMaxGhosts][64] + lane
    uint32_t n;
    float min_tfar;
    float tie_t;
    uint32_t tie_id;
    __device__ __forceinline__ void add(float tfar, uint32_t gid) {
        if (!(tfar >= min_tfar)) return;
        if (n < (uint32_t)kGrtMaxGhosts) id[n * 64] = gid;
        ++n;
    }
};

// one optixTrace: the (up to) 16 nearest candidates with t in (tmin, tmax), ascending in (t, particle)
struct TraceCounters {
    uint32_t nodes = 0, leaf_tests = 0, inserts = 0, rounds = 0, processed = 0, rej[4] = {0, 0, 0, 0}, wave_leaves = 0, wave_slab = 0, wave_insert = 0, batch_loads = 0;
    unsigned long long phase[3] = {0ull, 0ull, 0ull}, test_ticks = 0ull;   // (test_ticks: the part of the scans spent in the candidate tests' work loops)   // instrumented launches: 10 ns ticks a packet's wave spent in the rounds' scans / the hit log / the per-hit evaluation
};

// one optixTrace for every ray of the wave: the (up to) 16 nearest candidates with t in (tmin, tmax), ascending in
// (t, particle), per lane.  PACKET traversal: the 64 rays of an 8x8 pixel block walk the tree TOGETHER — one wave-uniform
// stack (LDS, 64 words per wave), node and proxy records fetched once per wave through the scalar path, every lane
// tests its own ray against the two child boxes, and a child is entered when ANY lane needs it.  Primary rays of a
// block are coherent, so the union of their paths is barely larger than one ray's path: the node fetches, the stack
// traffic and the divergence of 64 independent walks collapse into one.  Pruning stays per lane (its own interval and
// its own current 16th-nearest distance); lanes that are done (`active` false) just ride along.
template <bool COUNT, int G = kGrtMaxHits, bool GHOST = false>
__device__ __forceinline__ void trace_round(const GrtBvh& bvh, const RayW& r, float tmin, float tmax, bool active, int lane,
                                            uint32_t* __restrict__ stack /* [kGrtStackDepth] per wave */, HitBufferT<G>& buf, TraceCounters& tc,
                                            GhostLog* ghosts = nullptr) {
    buf.clear();
    if (COUNT && active) tc.rounds++;
    if (!__any(active)) return;
    int sp = 0;
    uint32_t cur = 0;  // root
    bool have = true;
    while (true) {
        if (!have) {
            if (sp == 0) break;
            cur = stack[--sp];
        }
        have = false;
        cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
        if (COUNT && lane == 0) tc.nodes++;
        // `cur` is wave-uniform and the tree is read-only while rays are traced: the constant address space makes this ONE
        // scalar fetch per wave (node in SGPRs) instead of four vector loads that occupy 16 VGPRs per lane
        const cfloat4* nq = reinterpret_cast<const cfloat4*>(reinterpret_cast<uintptr_t>(&bvh.nodes[cur]));
        const float4 q0 = ld4(nq, 0), q1 = ld4(nq, 1), q2 = ld4(nq, 2), q3 = ld4(nq, 3);   // q3 = {c0, c1, slack0, slack1}
        const uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
        float tn0, tf0, tn1, tf1;
        bool ok0, ok1;
        boxes_hit(q0, q1, q2, r, tn0, tf0, tn1, tf1, ok0, ok1);
        const float bound = fminf(tmax, buf.t[G - 1]);
        // (GHOST: the walk must also reach the proxies whose box the ray has left before tmin, as far back as the ghosts that are kept)
        const float exit_min = GHOST ? fminf(tmin, ghosts->min_tfar) : tmin;
        const bool h0 = active && (c0 != kGrtNoChild) && ok0 && (tf0 >= exit_min) && (tn0 <= tmax) && (tn0 - q3.z <= bound);
        const bool h1 = active && (c1 != kGrtNoChild) && ok1 && (tf1 >= exit_min) && (tn1 <= tmax) && (tn1 - q3.w <= bound);
        bool a0 = __any(h0), a1 = __any(h1);
        // leaves are tested on the spot, by the lanes whose ray touches the leaf's box
        auto leaf = [&](uint32_t c, bool h) {
            const uint32_t id = c & ~kGrtLeafBit;
            if (h) {
                const Cand cd = GHOST ? candidate_uniform<true>(bvh.inst + 12 * (size_t)id, r, ghosts->tie_t, buf.t[G - 1], id, buf.id[G - 1])
                                      : candidate_uniform(bvh.inst + 12 * (size_t)id, r, tmin, buf.t[G - 1], id, buf.id[G - 1]);
                const bool reach = cd.ok && (cd.t < tmax) && (cd.tnear <= tmax) && hit_less(cd.t, id, buf.t[G - 1], buf.id[G - 1]);
                const bool in_range = reach && (cd.t > tmin);
                const bool ins = in_range && (cd.tfar >= tmin);
                if (GHOST && ((in_range && !ins) || (reach && !in_range && hit_less(ghosts->tie_t, ghosts->tie_id, cd.t, id)))) ghosts->add(cd.tfar, id);
                if (COUNT) {   // per lane: outcome of the test; per wave (first active lane): did the box part / the insert chain run at all
                    tc.leaf_tests++; tc.rej[cd.ok ? 0 : cd.why]++;
                    const unsigned long long ms = __ballot(cd.ok || cd.why != 1), mi = __ballot(ins);
                    if (ms && lane == __ffsll((long long)ms) - 1) tc.wave_slab++;
                    if (mi && lane == __ffsll((long long)mi) - 1) tc.wave_insert++;
                }
                if (ins) {
                    buf.insert(cd.t, id);
                    if (COUNT) tc.inserts++;
                }
            }
            if (COUNT && lane == 0) tc.wave_leaves++;
        };
        if (a0 && (c0 & kGrtLeafBit)) { leaf(c0, h0); a0 = false; }
        if (a1 && (c1 & kGrtLeafBit)) { leaf(c1, h1); a1 = false; }
        if (a0 && a1) {  // enter the child most lanes reach first, keep the other
            const int v0 = __popcll(__ballot(h0 && (!h1 || tn0 <= tn1))), v1 = __popcll(__ballot(h1 && (!h0 || tn1 < tn0)));
            const bool first0 = v0 >= v1;
            if (lane == 0) stack[sp] = first0 ? c1 : c0;
            sp++;
            cur = first0 ? c0 : c1;
            have = true;
        } else if (a0 || a1) {
            cur = a0 ? c0 : c1;
            have = true;
        }
    }
}

// One optixTrace per LANE (round 6): every ray walks the tree on its own - its own current node, its own stack (LDS, `depth` words per
// lane at `lstack[k * 64 + lane]`), the nearer child first - for rays that do NOT share their path: the hybrid tracer's bounced segments
// (mirror, glass and PBR bounces leave a packet's 64 rays with 64 origins and directions).  The packet walk above visits the UNION of the
// 64 paths one node after the other (thousands of ~0.66 us steps for a wave with a single live ray as for 64 incoherent ones); here a
// step serves 64 different nodes at once and a round takes as many steps as its longest single path (a few hundred).  Same candidate test,
// same per-lane buffers ordered by (distance, particle): the result of a round does not depend on the order in which candidates arrive, so
// it equals the packet walk's bit for bit.  Pruning: the packet walk's own per-lane conditions, against the lane's current bound.
// Returns false if some lane's stack overflowed (the caller repeats the round with the packet walk; not observed on the bench scenes).
template <int G>
__device__ __forceinline__ bool trace_round_lanes(const GrtBvh& bvh, const RayW& r, float tmin, float tmax, bool active, int lane,
                                                  uint32_t* __restrict__ lstack, int depth, HitBufferT<G>& buf) {
    buf.clear();
    if (!__any(active)) return true;
    constexpr uint32_t kDone = 0xFFFFFFFEu;
    uint32_t cur = active ? 0u : kDone;   // root
    int sp = 0;
    bool overflow = false;
    const float4* nodes4 = reinterpret_cast<const float4*>(bvh.nodes);
    while (__any(cur != kDone)) {
        if (cur != kDone) {
            const float4 q0 = nodes4[4 * (size_t)cur], q1 = nodes4[4 * (size_t)cur + 1], q2 = nodes4[4 * (size_t)cur + 2], q3 = nodes4[4 * (size_t)cur + 3];
            const uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
            float tn0, tf0, tn1, tf1;
            bool ok0, ok1;
            boxes_hit(q0, q1, q2, r, tn0, tf0, tn1, tf1, ok0, ok1);
            const float bound = fminf(tmax, buf.t[G - 1]);
            bool h0 = (c0 != kGrtNoChild) && ok0 && (tf0 >= tmin) && (tn0 <= tmax) && (tn0 - q3.z <= bound);
            bool h1 = (c1 != kGrtNoChild) && ok1 && (tf1 >= tmin) && (tn1 <= tmax) && (tn1 - q3.w <= bound);
            const bool l0 = h0 && (c0 & kGrtLeafBit), l1 = h1 && (c1 & kGrtLeafBit);
            if (l0 || l1) {   // at most one pass per leaf child; the two leaves of a node are tested one after the other (the nearer box first)
                const bool first1 = l1 && (!l0 || tn1 < tn0);
#pragma unroll 1
                for (int k = 0; k < 2; ++k) {
                    const bool take1 = (k == 0) ? first1 : !first1;
                    const bool doit = take1 ? l1 : l0;
                    if (doit) {
                        const uint32_t id = (take1 ? c1 : c0) & ~kGrtLeafBit;
                        const Cand cd = candidate(bvh.inst + 12 * (size_t)id, r, tmin, buf.t[G - 1], id, buf.id[G - 1]);
                        const bool reach = cd.ok && (cd.t < tmax) && (cd.tnear <= tmax) && hit_less(cd.t, id, buf.t[G - 1], buf.id[G - 1]);
                        if (reach && (cd.t > tmin) && (cd.tfar >= tmin)) buf.insert(cd.t, id);
                    }
                }
                h0 = h0 && !l0; h1 = h1 && !l1;
            }
            if (h0 && h1) {   // both inner children: the nearer one now, the other parked
                const bool near0 = tn0 <= tn1;
                if (sp < depth) lstack[sp * 64 + lane] = near0 ? c1 : c0; else overflow = true;
                sp += sp < depth ? 1 : 0;
                cur = near0 ? c0 : c1;
            } else if (h0 || h1) {
                cur = h0 ? c0 : c1;
            } else if (sp > 0) {
                cur = lstack[--sp * 64 + lane];
            } else {
                cur = kDone;
            }
        }
    }
    return !__any(overflow);
}

// Tight bounds of the hit distance of one particle for the rays of ONE packet (cone axis `k`, half angle theta).  With x = o + t d - mu
// the closest-approach point, y = W x and u = W dh (dh the unit direction): y is perpendicular to u (t minimises |W x|) and |y| <= sqrt 3
// (the ray touches the proxy box), and t |d| = v.dh + x.dh with x.dh = y.(S e), e = R^T dh, u = S^-1 e, (S e).u = 1, hence
//     |x.dh| <= sqrt 3 sqrt(|S e|^2 - 1 / |S^-1 e|^2)      (zero for a sphere: its hit distance IS v.dh)
// Over the cone, v.dh = |v| cos(phi) with phi within theta of the angle between v and the axis; |S e| is kmax-Lipschitz and |S^-1 e|
// (1/kmin)-Lipschitz in dh.  (S e)_i = kscl_i^2 (W_i . dh), (S^-1 e)_i = W_i . dh.
// (hardware reciprocal / square root, 1 ulp each: the bounds are conservative by their margins - 1e-5 relative on every norm, 3e-6 L and
// 6e-6 on the angular part, which cover a dozen such roundings - not by correctly rounded operations; the IEEE sequences were 200
// instructions per staged entry, 7 % of the forward's issue time)
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ void packet_bounds(const GrtCone& k, f3 v, float L2, const float4& a, const float4& b, float w22, float key, float ub,
                                              float dmin, float dmax, float& lo, float& hi) {
    lo = key; hi = ub;
    if (k.cos_t <= -1.f) return;   // cone of everything
    const float L = fast_sqrt(L2);
    const float k0 = fast_rcp(a.x * a.x + a.y * a.y + a.z * a.z), k1 = fast_rcp(a.w * a.w + b.x * b.x + b.y * b.y), k2 = fast_rcp(b.z * b.z + b.w * b.w + w22 * w22);   // kscl^2
    const float kmax = fast_sqrt(fmaxf(k0, fmaxf(k1, k2))), ikmin = __builtin_amdgcn_rsqf(fminf(k0, fminf(k1, k2)));
    const float w0 = a.x * k.ax + a.y * k.ay + a.z * k.az, w1 = a.w * k.ax + b.x * k.ay + b.y * k.az, w2 = b.z * k.ax + b.w * k.ay + w22 * k.az;
    const float s0 = k0 * w0, s1 = k1 * w1, s2 = k2 * w2;
    const float chord = fast_sqrt(fmaxf(0.f, 2.f * (1.f - k.cos_t))) * 1.00001f;
    const float se = (fast_sqrt(s0 * s0 + s1 * s1 + s2 * s2) + kmax * chord) * 1.00001f;       // >= |S e| over the cone
    const float sie = (fast_sqrt(w0 * w0 + w1 * w1 + w2 * w2) + chord * ikmin) * 1.00001f;      // >= |S^-1 e| over the cone
    const float h = 1.7320509f * fast_sqrt(fmaxf(0.f, se * se - (1.f - 1e-5f) * fast_rcp(sie * sie))) * 1.00002f + 3e-6f * L + 1e-30f;
    const float ca = L > 0.f ? fminf(1.f, fmaxf(-1.f, (v.x * k.ax + v.y * k.ay + v.z * k.az) * fast_rcp(L))) : 1.f;
    const float sa = fast_sqrt(fmaxf(0.f, 1.f - ca * ca));
    const float cmax = (ca >= k.cos_t) ? 1.f : fminf(1.f, ca * k.cos_t + sa * k.sin_t + 6e-6f);     // cos of (alpha - theta), or 1 inside the cone
    const float cmin = (ca <= -k.cos_t) ? -1.f : fmaxf(-1.f, ca * k.cos_t - sa * k.sin_t - 6e-6f);  // cos of (alpha + theta), or -1 past pi
    const float tl = L * cmin - h, th = L * cmax + h;   // bounds of t |d|
    const float l2 = tl > 0.f ? (tl * fast_rcp(dmax)) * (1.f - 4e-6f) : 0.f;
    const float h2 = th > 0.f ? (th * fast_rcp(dmin)) * (1.f + 4e-6f) + 1e-30f : 0.f;
    lo = fmaxf(lo, l2);
    hi = fminf(hi, h2);
}
// ---------------------------------------------------------------------------------------------
// packet lists (GrtLists): one k = 16 trace round as a scan of a window of the packet's candidate list
// ---------------------------------------------------------------------------------------------
// min / max over the wave of any floats: the DPP modifier rides on the min / max instruction itself (v_min_f32_dpp: four steps inside the
// rows of 16, then row_bcast:15 / row_bcast:31 carry the rows' results down to lane 63) - 6 instructions + one v_readlane_b32 where the
// compiler's form of fminf(v, dpp(v)) is mov_dpp + two canonicalisations + min per step and four readlanes (25 issue slots: a tenth of
// the forward's VALU time went into the two reductions of every first test).  IEEE mode: a NaN operand loses, like fminf / fmaxf.
// (the hazard between a VALU write and a DPP read of the same register - two wait states - is covered by the s_nop in front of each step:
// the compiler does not look into inline asm)
#define GRT_DPP_STEP(op, ctl) "s_nop 1\n\t" op " %0, %0, %0 " ctl " row_mask:0xf bank_mask:0xf\n\t"
template <bool MAX>
__device__ __forceinline__ float wave_extreme(float v) {
    if (MAX) {
        asm(GRT_DPP_STEP("v_max_f32_dpp", "quad_perm:[1,0,3,2]") GRT_DPP_STEP("v_max_f32_dpp", "quad_perm:[2,3,0,1]") GRT_DPP_STEP("v_max_f32_dpp", "row_half_mirror")
            GRT_DPP_STEP("v_max_f32_dpp", "row_mirror")
            "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
            "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 0"
            : "+v"(v));
    } else {
        asm(GRT_DPP_STEP("v_min_f32_dpp", "quad_perm:[1,0,3,2]") GRT_DPP_STEP("v_min_f32_dpp", "quad_perm:[2,3,0,1]") GRT_DPP_STEP("v_min_f32_dpp", "row_half_mirror")
            GRT_DPP_STEP("v_min_f32_dpp", "row_mirror")
            "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
            "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 0"
            : "+v"(v));
    }
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_nonneg(float v) { return wave_extreme<true>(v); }
// the minimum of `lo` and the maximum of `hi` over the wave in one interleaved sequence (each chain's steps sit in the other's wait states)
#define GRT_DPP_PAIR(ctl) "v_min_f32_dpp %0, %0, %0 " ctl "\n\tv_max_f32_dpp %1, %1, %1 " ctl "\n\ts_nop 0\n\t"
__device__ __forceinline__ void wave_min_max(float lo, float hi, float& lo_all, float& hi_all) {
    asm("s_nop 1\n\t" GRT_DPP_PAIR("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") GRT_DPP_PAIR("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
        GRT_DPP_PAIR("row_half_mirror row_mask:0xf bank_mask:0xf") GRT_DPP_PAIR("row_mirror row_mask:0xf bank_mask:0xf")
        GRT_DPP_PAIR("row_bcast:15 row_mask:0xa bank_mask:0xf") GRT_DPP_PAIR("row_bcast:31 row_mask:0xc bank_mask:0xf")
        : "+v"(lo), "+v"(hi));
    lo_all = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lo), 63));
    hi_all = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hi), 63));
}
// half extents of a proxy's vertices along the axes of its frame (the unit cube: 1): the polyhedron's table entry; GRUT_PRIM_TRIHEXA: proxy
// 3 p + j is the rhombus in plane j - flat along axis j, sqrt 2 along the other two
__device__ __forceinline__ void proxy_extents(int prim, uint32_t id, float (&ext)[3]) {
    if (prim == GRUT_PRIM_TRIHEXA) {
        const uint32_t plane = id % 3u;
        ext[0] = plane == 0u ? 0.f : 1.4142135381698608f; ext[1] = plane == 1u ? 0.f : 1.4142135381698608f; ext[2] = plane == 2u ? 0.f : 1.4142135381698608f;
    } else if (prim == GRUT_PRIM_SPHERE) {
        ext[0] = ext[1] = ext[2] = 1.f;   // (the unit sphere of the frame scaled by the radius)
    } else {
        ext[0] = kGrtPolyhedra[prim].ext[0]; ext[1] = kGrtPolyhedra[prim].ext[1]; ext[2] = kGrtPolyhedra[prim].ext[2];
    }
}
struct ListEntry {
    float4 a, b, e;    // proxy record {W rows, W (o - mu)}
    uint32_t id;       // particle
    float lo, hi, key; // hit-distance bounds for THIS packet's rays, and the list's sort key (a bound for every ray of the frame)
    bool fresh;        // the bounds are the geometric ones: no round of this packet has tested the entry yet
};
// one entry per lane: the particle's records are gathered and its packet-specific bounds computed on the spot (one lane per entry:
// a few dozen operations per 64 entries of wave time)
__device__ __forceinline__ ListEntry load_list_entry(const GrtLists& L, const GrtCone& cone, float dmin, float dmax, uint32_t e, uint32_t end, int prim = GRUT_PRIM_INSTANCES,
                                                     const float* __restrict__ box8 = nullptr) {
    ListEntry x;
    x.a = x.b = x.e = make_float4(0.f, 0.f, 0.f, 0.f);
    x.id = 0xFFFFFFFFu; x.lo = 3.0e38f; x.hi = -3.0e38f; x.key = 3.0e38f;   // dead for every ray, beyond every bound
    x.fresh = false;
    if (e < end) {
        // (device-scope accesses: the flag and the interval were written by this packet's own wave in an earlier round — never serve them
        // from a stale L1 line)
        const uint32_t word = __hip_atomic_load(const_cast<uint32_t*>(L.entries) + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x.id = word;
        if (word != 0xFFFFFFFFu) {
            x.id = word & ~kGrtEntryRefined;
            const float4* rec = reinterpret_cast<const float4*>(L.inst_rel) + 4 * (size_t)x.id;   // one 64-byte line per entry
            x.a = rec[0]; x.b = rec[1]; x.e = rec[2];
            const float4 vk = rec[3];
            const f3 v = mk3(vk.x, vk.y, vk.z);
            x.key = vk.w;
            if (word & kGrtEntryRefined) {   // the packet's own rays have been through this entry (see list_round)
                const unsigned long long raw = __hip_atomic_load(reinterpret_cast<unsigned long long*>(L.bounds + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                x.lo = __uint_as_float((uint32_t)raw); x.hi = __uint_as_float((uint32_t)(raw >> 32));
            } else if (prim == GRUT_PRIM_CUSTOM) {
                // custom primitives: an accepted hit lies within 3 sigma of the SCALE frame, |x* - mu| < 3 kmax / ks (bin_particle): the hit
                // distance within that sphere of the centre's distance, until the packet's first test refines it
                const float kmax = fast_rcp(fast_sqrt(fminf(x.a.x * x.a.x + x.a.y * x.a.y + x.a.z * x.a.z, fminf(x.a.w * x.a.w + x.b.x * x.b.x + x.b.y * x.b.y,
                                                                                                                 x.b.z * x.b.z + x.b.w * x.b.w + x.e.x * x.e.x))));
                const float Rt = 3.f * kmax * fast_rcp(fast_sqrt(box8[8 * (size_t)x.id + 6])) * 1.0002f;
                x.lo = vk.w;
                x.hi = ((sqrtf(dot(v, v)) * (1.f + 2e-6f) + Rt) / dmin) * (1.f + 2e-6f) + 1e-30f;
                x.fresh = true;
            } else if (prim != GRUT_PRIM_INSTANCES) {
                // mesh proxies: until the packet's first test refines it, the entry distance lies within the bounding sphere of the
                // polyhedron's box around the centre's distance (the key is that sphere's near end over the frame's direction lengths)
                float ext[3];
                proxy_extents(prim, x.id, ext);
                const float k0 = ext[0] * ext[0] / (x.a.x * x.a.x + x.a.y * x.a.y + x.a.z * x.a.z),
                            k1 = ext[1] * ext[1] / (x.a.w * x.a.w + x.b.x * x.b.x + x.b.y * x.b.y),
                            k2 = ext[2] * ext[2] / (x.b.z * x.b.z + x.b.w * x.b.w + x.e.x * x.e.x);
                x.lo = vk.w;
                x.hi = ((sqrtf(dot(v, v)) * (1.f + 2e-6f) + sqrtf(k0 + k1 + k2) * 1.00002f) / dmin) * (1.f + 2e-6f) + 1e-30f;
                x.fresh = true;
            } else {
                packet_bounds(cone, v, dot(v, v), x.a, x.b, x.e.x, vk.w, 3.0e38f, dmin, dmax, x.lo, x.hi);
                x.fresh = true;
            }
        }
    }
    return x;
}
// One optixTrace for every ray of the wave out of the packet's list [.., le) (ascending key): the 16 nearest candidates with t in
// (tmin, tmax), per lane, with the candidate arithmetic and the buffers of trace_round — same sets, same order.  Entries are fetched
// 64 at a time, one per lane (record gathered by particle), and parked in LDS; ONE vector compare per batch then tells which of them
// can matter to some ray of the wave now — [lo, hi] reaches past the rays' last hit distances (hi >= min over lanes of tmin) and not
// beyond every lane's current 16th-nearest distance (lo <= max over lanes of the bound) — and only those are tested, by all lanes,
// against wave-uniform LDS reads (fetching the records through the scalar cache instead, one request ahead, was measured slower:
// 22.6 vs 20.0 ms).  The scan starts at `start` (everything before it is dead for every ray for good: hi < the rays' last hit
// distances, which only grow) and ends with the first key beyond the bound (keys are lower bounds of lo: nothing that follows can
// enter a buffer).
// REFINE (the forward): the first time an entry is tested, every running ray takes the box test, and the smallest and largest hit
// distance over the rays that MEET the proxy are kept (GrtLists::bounds; an empty interval when none does) and replace the geometric
// [lo, hi] from then on — in every later round of this launch and in the backward's
// re-derivations.  Rays only ever stop, and a ray needs an entry no later than the round whose window holds its hit distance, so the
// rays that were running at the first test are all the rays that can ever want the entry: the interval is exact for them.  (A needle's
// geometric interval spans its whole length; the window of a round slides through it in up to ten rounds, each of which tested the
// entry against all 64 rays for nothing.)
// Two passes per round (REFINE): rays advance by about the same distance every round, so the round first scans only the entries whose
// (refined) interval starts within the previous round's reach of the slowest ray (`span`, kept by the caller); when that fills
// every running ray's buffer below the mark — the usual case — the entries beyond it are never tested, instead of being inserted into
// the still empty buffers and pushed out again by the nearer ones that follow (140 M insertions for 43 M processed hits).  Otherwise a
// second pass goes through the deferred entries with whatever bound the first pass left.  Entries refined during a round carry their
// flag only from the round's end (`pending`, LDS): in the second pass an entry without the flag has been tested by the first.
constexpr uint32_t kGrtPendingCap = 512;
#ifndef GRT_LIST_ILP
#define GRT_LIST_ILP 1   // entries tested per iteration of list_round's work loop (2: two independent candidate chains)
#endif
template <bool COUNT, int G, bool GHOST = false, bool REFINE = false>
__device__ __forceinline__ void list_round(const GrtLists& L, const GrtCone& cone, float dmin, float dmax, uint32_t le, uint32_t& start, const RayW& r,
                                           float tmin, float tmax, bool active, int lane, float4* __restrict__ s_ent /* [64][3] */, HitBufferT<G>& buf,
                                           TraceCounters& tc, GhostLog* ghosts = nullptr, uint32_t* __restrict__ pending = nullptr /* LDS [kGrtPendingCap] */,
                                           float* span = nullptr, float mark_factor = 1.f) {
    buf.clear();
    if (COUNT && active) tc.rounds++;
    if (!__any(active)) return;
    const float wmin_tmin = wave_extreme<false>(active ? tmin : 3.0e38f);
    float wmax_bound = wave_max_nonneg(active ? tmax : -1.f);   // no lane holds 16 candidates yet
    const float mark = (REFINE && *span < 1.0e37f) ? wmin_tmin + mark_factor * *span : 3.0e38f;
    uint32_t n_pending = 0u;
    bool deferred = false;
    for (int pass = 0; pass < (REFINE ? 2 : 1); ++pass) {
        if (pass == 1 && !(deferred && wmax_bound > mark)) break;
        bool seen_live = pass == 1;   // (the scan start only moves in the first pass)
        uint32_t base = start & ~63u;
        ListEntry nxt = load_list_entry(L, cone, dmin, dmax, base + lane, le, r.prim, r.box8);
        if (COUNT && lane == 0) tc.batch_loads++;
        while (base < le) {
            s_ent[lane * 3 + 0] = nxt.a; s_ent[lane * 3 + 1] = nxt.b; s_ent[lane * 3 + 2] = nxt.e;
            const uint32_t my_id = nxt.id;
            const float my_lo = nxt.lo, my_hi = nxt.hi, my_key = nxt.key;
            const unsigned long long fresh = REFINE ? __ballot(nxt.fresh) : 0ull;
            __syncthreads();   // single-wave workgroup: orders the LDS hand-off
            const uint32_t bend = min(le, base + 64u);
            nxt = load_list_entry(L, cone, dmin, dmax, bend + lane, le, r.prim, r.box8);   // the following batch travels while this one is tested
            if (COUNT && lane == 0 && bend < le) tc.batch_loads++;
            const bool mine = (base + (uint32_t)lane >= start) && (base + (uint32_t)lane < bend);
            const unsigned long long beyond = __ballot(mine && my_key > wmax_bound);           // a suffix of the batch (keys ascend)
            const unsigned long long dead = __ballot(!mine || my_hi < wmin_tmin);              // behind every ray's last hit (or not in the scan)
            unsigned long long live = __ballot(mine && !(my_hi < wmin_tmin) && !(my_lo > wmax_bound)) & (beyond ? ((1ull << (__ffsll((long long)beyond) - 1)) - 1ull) : ~0ull);
            if (REFINE) {
                const unsigned long long far = __ballot(my_lo > mark) & ~fresh;   // (an entry without the flag is tested by the first pass)
                if (pass == 0) { deferred = deferred || (live & far) != 0ull; live &= ~far; }
                else live &= far;
            }
            if (!seen_live) {   // the scan start moves past the leading entries that are dead for good
                const unsigned long long alive = ~dead;
                const uint32_t lead = alive ? (uint32_t)(__ffsll((long long)alive) - 1) : 64u;
                start = max(start, min(base + lead, bend));
                seen_live = alive != 0ull;
            }
            int tested = 0;
            const unsigned long long tt0 = (COUNT && live) ? wall_clock64() : 0ull;
            const bool timed_tests = COUNT && live;
#if GRT_LIST_ILP == 2
            // Two live entries per iteration: their candidate tests are independent instruction chains (the second one's early-out uses
            // the bound BEFORE the first one's insertion - a superset; `reach` re-checks against the buffer as it then stands, so
            // decisions, buffers and ghosts are those of the one-by-one loop).  A packet's wave is latency-bound when its SIMD is not
            // full - the last 45 % of the launch's span - and the two chains fill each other's issue gaps.
            while (live) {
                const int j0 = __ffsll((long long)live) - 1;
                live &= live - 1;
                const bool two = live != 0ull;
                const int j1 = two ? __ffsll((long long)live) - 1 : j0;
                live &= live - 1;   // (0 & -1 = 0 when there was no second entry)
                const uint32_t id0 = (uint32_t)__builtin_amdgcn_readlane((int)my_id, j0), id1 = (uint32_t)__builtin_amdgcn_readlane((int)my_id, j1);
                if (COUNT && lane == 0) tc.wave_leaves += two ? 2u : 1u;
                const bool ft0 = REFINE && pass == 0 && ((fresh >> j0) & 1ull) && n_pending < kGrtPendingCap;
                const bool ft1 = two && REFINE && pass == 0 && ((fresh >> j1) & 1ull) && n_pending + (ft0 ? 1u : 0u) < kGrtPendingCap;
                float lo0 = 3.0e38f, hi0 = -3.0e38f, lo1 = 3.0e38f, hi1 = -3.0e38f;
                if (active) {
                    const float t_lo = GHOST ? ghosts->tie_t : tmin;
                    const Cand c0 = GHOST ? candidate_abe<true, true>(s_ent[j0 * 3], s_ent[j0 * 3 + 1], s_ent[j0 * 3 + 2], r, t_lo, buf.t[G - 1], id0, buf.id[G - 1], ft0, true)
                                          : candidate_abe<true>(s_ent[j0 * 3], s_ent[j0 * 3 + 1], s_ent[j0 * 3 + 2], r, t_lo, buf.t[G - 1], id0, buf.id[G - 1], ft0, true);
                    Cand c1 = GHOST ? candidate_abe<true, true>(s_ent[j1 * 3], s_ent[j1 * 3 + 1], s_ent[j1 * 3 + 2], r, t_lo, buf.t[G - 1], id1, buf.id[G - 1], ft1, true)
                                    : candidate_abe<true>(s_ent[j1 * 3], s_ent[j1 * 3 + 1], s_ent[j1 * 3 + 2], r, t_lo, buf.t[G - 1], id1, buf.id[G - 1], ft1, true);
                    if (!two) { c1.ok = false; c1.box = false; }
                    auto take = [&](const Cand& cd, uint32_t id, float& tl, float& th) {
                        if (REFINE && cd.box) { tl = cd.t; th = cd.t; }
                        const bool reach = cd.ok && (cd.t < tmax) && (cd.tnear <= tmax) && hit_less(cd.t, id, buf.t[G - 1], buf.id[G - 1]);
                        const bool in_range = reach && (cd.t > tmin);
                        const bool ins = in_range && (cd.tfar >= tmin);
                        if (GHOST && ((in_range && !ins) || (reach && !in_range && hit_less(ghosts->tie_t, ghosts->tie_id, cd.t, id)))) ghosts->add(cd.tfar, id);
                        if (COUNT) {
                            tc.leaf_tests++; tc.rej[cd.ok ? 0 : cd.why]++;
                            const unsigned long long ms = __ballot(cd.ok || cd.why != 1), mi = __ballot(ins);
                            if (ms && lane == __ffsll((long long)ms) - 1) tc.wave_slab++;
                            if (mi && lane == __ffsll((long long)mi) - 1) tc.wave_insert++;
                        }
                        if (ins) {
                            buf.insert(cd.t, id);
                            if (COUNT) tc.inserts++;
                        }
                    };
                    take(c0, id0, lo0, hi0);
                    if (two) take(c1, id1, lo1, hi1);
                }
                if (ft0 || ft1) {
                    float l0, h0, l1, h1;
                    wave_min_max(lo0, hi0, l0, h0);
                    wave_min_max(lo1, hi1, l1, h1);
                    if (ft0) {
                        if (lane == 0) {
                            __hip_atomic_store(reinterpret_cast<unsigned long long*>(L.bounds + base + (uint32_t)j0),
                                               (unsigned long long)__float_as_uint(l0) | ((unsigned long long)__float_as_uint(h0) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            pending[n_pending] = base + (uint32_t)j0;
                        }
                        ++n_pending;
                    }
                    if (ft1) {
                        if (lane == 0) {
                            __hip_atomic_store(reinterpret_cast<unsigned long long*>(L.bounds + base + (uint32_t)j1),
                                               (unsigned long long)__float_as_uint(l1) | ((unsigned long long)__float_as_uint(h1) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            pending[n_pending] = base + (uint32_t)j1;
                        }
                        ++n_pending;
                    }
                }
                tested += 2;
                if ((tested & 7) == 0 && live) {
                    wmax_bound = wave_max_nonneg(active ? fminf(tmax, buf.t[G - 1]) : -1.f);
                    live &= __ballot(!(my_lo > wmax_bound));
                }
            }
#else
            while (live) {
                const int j = __ffsll((long long)live) - 1;
                live &= live - 1;
                const uint32_t id = (uint32_t)__builtin_amdgcn_readlane((int)my_id, j);
                if (COUNT && lane == 0) tc.wave_leaves++;
                float t_mine_lo = 3.0e38f, t_mine_hi = -3.0e38f;
                const bool first_test = REFINE && pass == 0 && ((fresh >> j) & 1ull) && n_pending < kGrtPendingCap;
                if (active) {
                    const Cand cd = GHOST ? candidate_abe<true, true>(s_ent[j * 3], s_ent[j * 3 + 1], s_ent[j * 3 + 2], r, ghosts->tie_t, buf.t[G - 1], id, buf.id[G - 1], first_test, true)
                                          : candidate_abe<true>(s_ent[j * 3], s_ent[j * 3 + 1], s_ent[j * 3 + 2], r, tmin, buf.t[G - 1], id, buf.id[G - 1], first_test, true);
                    if (REFINE && cd.box) { t_mine_lo = cd.t; t_mine_hi = cd.t; }
                    const bool reach = cd.ok && (cd.t < tmax) && (cd.tnear <= tmax) && hit_less(cd.t, id, buf.t[G - 1], buf.id[G - 1]);
                    const bool in_range = reach && (cd.t > tmin);
                    const bool ins = in_range && (cd.tfar >= tmin);
                    if (GHOST && ((in_range && !ins) || (reach && !in_range && hit_less(ghosts->tie_t, ghosts->tie_id, cd.t, id)))) ghosts->add(cd.tfar, id);
                    if (COUNT) {
                        tc.leaf_tests++; tc.rej[cd.ok ? 0 : cd.why]++;
                        const unsigned long long ms = __ballot(cd.ok || cd.why != 1), mi = __ballot(ins);
                        if (ms && lane == __ffsll((long long)ms) - 1) tc.wave_slab++;
                        if (mi && lane == __ffsll((long long)mi) - 1) tc.wave_insert++;
                    }
                    if (ins) {
                        buf.insert(cd.t, id);
                        if (COUNT) tc.inserts++;
                    }
                }
                if (first_test) {
                    // (a NaN distance fails every comparison of the candidate test: such a ray never wants the entry; fminf / fmaxf drop it)
                    float lo_all, hi_all;
                    wave_min_max(t_mine_lo, t_mine_hi, lo_all, hi_all);
                    if (lane == 0) {
                        __hip_atomic_store(reinterpret_cast<unsigned long long*>(L.bounds + base + (uint32_t)j),
                                           (unsigned long long)__float_as_uint(lo_all) | ((unsigned long long)__float_as_uint(hi_all) << 32), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                        pending[n_pending] = base + (uint32_t)j;
                    }
                    ++n_pending;
                }
                if ((++tested & 7) == 0 && live) {   // tighten the bound now and then: entries that fell beyond it leave the batch's work list
                    wmax_bound = wave_max_nonneg(active ? fminf(tmax, buf.t[G - 1]) : -1.f);
                    live &= __ballot(!(my_lo > wmax_bound));
                }
            }
#endif
            if (timed_tests) tc.test_ticks += wall_clock64() - tt0;
            __syncthreads();
            if (beyond) break;
            wmax_bound = wave_max_nonneg(active ? fminf(tmax, buf.t[G - 1]) : -1.f);
            base = bend;
        }
        wmax_bound = wave_max_nonneg(active ? fminf(tmax, buf.t[G - 1]) : -1.f);
    }
    if (REFINE) {
        __syncthreads();
        for (uint32_t i = (uint32_t)lane; i < n_pending; i += 64u) {   // the round's refined entries get their flag
            uint32_t* w = const_cast<uint32_t*>(L.entries) + pending[i];
            __hip_atomic_store(w, __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | kGrtEntryRefined, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // what the next round can expect: the farthest 16th candidate of the rays that filled their buffers (the others are done)
        const float reach_full = wave_max_nonneg((active && buf.id[G - 1] != 0xFFFFFFFFu) ? buf.t[G - 1] : -1.f);
        *span = reach_full >= 0.f ? fmaxf(0.f, reach_full - wmin_tmin) : 3.0e38f;
    }
}

// ---------------------------------------------------------------------------------------------
// per-hit math (gaussianParticles.cuh:337-405, :468-731): scalar per ray, SH evaluated along the ray
// ---------------------------------------------------------------------------------------------
struct Particle {
    f3 pos, scl;
    float4 quat;
    m3 rotT;
    float density;
};
template <typename Q>
__device__ __forceinline__ Particle load_particle_q(const Q* __restrict__ rec) {
    const float4 a = ld4(rec, 0), q = ld4(rec, 1), s = ld4(rec, 2);
    Particle p;
    p.pos = mk3(a.x, a.y, a.z); p.density = a.w; p.quat = q; p.scl = mk3(s.x, s.y, s.z);
    p.rotT = quat_wxyz_to_rotT(q.x, q.y, q.z, q.w);
    return p;
}
// wave-uniform particle of a read-only parameter array: one scalar fetch per wave
__device__ __forceinline__ Particle load_particle_uniform(const float4* density12, uint32_t id) {
    return load_particle_q(reinterpret_cast<const cfloat4*>(reinterpret_cast<uintptr_t>(density12 + 3 * (size_t)id)));
}
__device__ __forceinline__ Particle load_particle(const float4* __restrict__ density12, uint32_t id) {
    const float4 a = density12[3 * (size_t)id], q = density12[3 * (size_t)id + 1], s = density12[3 * (size_t)id + 2];
    Particle p;
    p.pos = mk3(a.x, a.y, a.z); p.density = a.w; p.quat = q; p.scl = mk3(s.x, s.y, s.z);
    p.rotT = quat_wxyz_to_rotT(q.x, q.y, q.z, q.w);
    return p;
}
__device__ __forceinline__ void sh_basis16(int deg, f3 d, float b[16]) {
    const float x = d.x, y = d.y, z = d.z;
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = 0.f;
    b[0] = 0.28209479177387814f;
    if (deg > 0) {
        const float c1 = 0.4886025119029199f;
        b[1] = -c1 * y; b[2] = c1 * z; b[3] = -c1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = 1.0925484305920792f * xy; b[5] = -1.0925484305920792f * yz; b[6] = 0.31539156525252005f * (2.f * zz - xx - yy);
            b[7] = -1.0925484305920792f * xz; b[8] = 0.5462742152960396f * (xx - yy);
            if (deg > 2) {
                b[9] = -0.5900435899266435f * y * (3.f * xx - yy);
                b[10] = 2.890611442640554f * xy * z;
                b[11] = -0.4570457994644658f * y * (4.f * zz - xx - yy);
                b[12] = 0.3731763325901154f * z * (2.f * zz - 3.f * xx - 3.f * yy);
                b[13] = -0.4570457994644658f * x * (4.f * zz - xx - yy);
                b[14] = 1.445305721320277f * z * (xx - yy);
                b[15] = -0.5900435899266435f * x * (xx - 3.f * yy);
            }
        }
    }
}
// the [H,W,3] integrated radiance: fp32, or IEEE half with FEATURE_OUTPUT_HALF (referenceSlangOptix.cu:183-184 / referenceSlangBwdOptix.cu:116-117)
__device__ __forceinline__ void store_radiance(const GrtTraceParams& P, float* __restrict__ out_rad, size_t pix, f3 rad) {
    if (P.out_half) {
        __half* h = reinterpret_cast<__half*>(out_rad) + 3 * pix;
        h[0] = __float2half(rad.x); h[1] = __float2half(rad.y); h[2] = __float2half(rad.z);
    } else {
        out_rad[3 * pix] = rad.x; out_rad[3 * pix + 1] = rad.y; out_rad[3 * pix + 2] = rad.z;
    }
}
__device__ __forceinline__ f3 load_radiance(const GrtTraceParams& P, const float* __restrict__ in_rad, size_t pix) {
    if (P.out_half) {
        const __half* h = reinterpret_cast<const __half*>(in_rad) + 3 * pix;
        return mk3(__half2float(h[0]), __half2float(h[1]), __half2float(h[2]));
    }
    return mk3(in_rad[3 * pix], in_rad[3 * pix + 1], in_rad[3 * pix + 2]);
}
// unclamped radiance along direction d
__device__ __forceinline__ f3 sh_radiance(const GrtTraceParams& P, const float* __restrict__ sph, uint32_t id, const float b[16]) {
    const float* c = sph + (size_t)id * 3 * P.ncoef;
    const int nact = min((P.sph_degree + 1) * (P.sph_degree + 1), P.ncoef);
    f3 rad = mk3(0.f, 0.f, 0.f);
    if (P.sph_half) {   // PARTICLE_FEATURE_HALF (optixTracer.cpp:54-56): half coefficients, fp32 arithmetic in the same order
        const __half* hc = reinterpret_cast<const __half*>(sph) + (size_t)id * 3 * P.ncoef;
        for (int k = 0; k < nact; ++k) {
            rad.x = fmaf(b[k], __half2float(hc[3 * k]), rad.x);
            rad.y = fmaf(b[k], __half2float(hc[3 * k + 1]), rad.y);
            rad.z = fmaf(b[k], __half2float(hc[3 * k + 2]), rad.z);
        }
        return rad + mk3(0.5f, 0.5f, 0.5f);
    }
    if (nact == 16 && (reinterpret_cast<uintptr_t>(sph) & 15u) == 0) {
        // the full degree-3 row is 192 bytes at a 16-byte-aligned address: twelve 16-byte requests per lane instead of 48 dword
        // loads (the per-hit gathers are the trace's vector-memory instruction stream), same order of accumulation
        const float4* q = reinterpret_cast<const float4*>(c);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 q0 = q[3 * g], q1 = q[3 * g + 1], q2 = q[3 * g + 2];   // coefficients 4g .. 4g+3, rgb interleaved
            rad.x = fmaf(b[4 * g], q0.x, rad.x); rad.y = fmaf(b[4 * g], q0.y, rad.y); rad.z = fmaf(b[4 * g], q0.z, rad.z);
            rad.x = fmaf(b[4 * g + 1], q0.w, rad.x); rad.y = fmaf(b[4 * g + 1], q1.x, rad.y); rad.z = fmaf(b[4 * g + 1], q1.y, rad.z);
            rad.x = fmaf(b[4 * g + 2], q1.z, rad.x); rad.y = fmaf(b[4 * g + 2], q1.w, rad.y); rad.z = fmaf(b[4 * g + 2], q2.x, rad.z);
            rad.x = fmaf(b[4 * g + 3], q2.y, rad.x); rad.y = fmaf(b[4 * g + 3], q2.z, rad.y); rad.z = fmaf(b[4 * g + 3], q2.w, rad.z);
        }
        return rad + mk3(0.5f, 0.5f, 0.5f);
    }
    for (int k = 0; k < nact; ++k) {
        rad.x = fmaf(b[k], c[3 * k], rad.x); rad.y = fmaf(b[k], c[3 * k + 1], rad.y); rad.z = fmaf(b[k], c[3 * k + 2], rad.z);
    }
    return rad + mk3(0.5f, 0.5f, 0.5f);
}

// the particle behind a proxy index (hit buffers, the log and the tree are keyed by proxy; GRUT_PRIM_TRIHEXA has three proxies per particle)
__device__ __forceinline__ uint32_t particle_of(const GrtTraceParams& P, uint32_t proxy) {
    return P.prim == GRUT_PRIM_TRIHEXA ? proxy / 3u : (P.prim == GRUT_PRIM_SPHERE ? proxy >> 1 : proxy);   // (GRUT_PRIM_SPHERE: the two roots)
}

struct HitGeom {
    f3 gposc, gposcr, gro, rdr, grdu, grd, gcrod, giscl;
    float gray, gres, galpha;
    bool accept;
};
template <int DEG>
__device__ __forceinline__ HitGeom hit_geometry(const GrtTraceParams& P, const Particle& p, const RayW& r) {
    HitGeom g;
    g.giscl = mk3(1.f / p.scl.x, 1.f / p.scl.y, 1.f / p.scl.z);
    g.gposc = r.o - p.pos;
    g.gposcr = mul_rows(p.rotT, g.gposc);
    g.gro = g.giscl * g.gposcr;
    g.rdr = mul_rows(p.rotT, r.d);
    g.grdu = g.giscl * g.rdr;
    const float l2 = dot(g.grdu, g.grdu);
    g.grd = l2 > 0.f ? g.grdu * (1.f / sqrtf(l2)) : g.grdu;
    if (P.prim == GRUT_PRIM_TRISURFEL) {
        // PipelineParameters::SurfelPrimitive (gaussianParticles.cuh:371-400): the response is evaluated where the ray crosses the particle's
        // z = 0 plane, gro + grd ghitT with ghitT = -gro.z / grd.z (component by component as the checker: (grd_k * -gro.z) / grd.z)
        const float nz = -g.gro.z;
        g.gcrod = mk3(g.gro.x + (g.grd.x * nz) / g.grd.z, g.gro.y + (g.grd.y * nz) / g.grd.z, g.gro.z + (g.grd.z * nz) / g.grd.z);
    } else {
        g.gcrod = cross(g.grd, g.gro);
    }
    g.gray = dot(g.gcrod, g.gcrod);
    g.gres = particle_response<DEG>(g.gray);
    g.galpha = fminf(P.max_alpha, g.gres * p.density);
    g.accept = (g.gres > P.min_response) && (g.galpha > P.min_alpha);
    return g;
}

// The two places where processHitBwd depends on the primitive (gaussianParticles.cuh:558-565 and :628-659), shared by the traversal
// backward and the log replay.  `surfel` is wave-uniform (GrtTraceParams::prim).
// d hitT -> d grd, d gro: the volumetric primitives measure the distance to the point of maximum response (pdot = -grd . gro), the
// surfel to the crossing of its z = 0 plane (pdot = -gro.z / grd.z, which depends on the z components only).  lead = gscl grdsRayHitGrd pdot.
__device__ __forceinline__ void surfel_hit_distance_grads(bool surfel, const HitGeom& g, f3 lead, float pdot, float grdScaledDot, f3& grdRayHitGrd,
                                                          f3& groRayHitGrd) {
    if (surfel) {
        grdRayHitGrd = lead - mk3(0.f, 0.f, (pdot / g.grd.z) * grdScaledDot);
        groRayHitGrd = mk3(0.f, 0.f, -grdScaledDot / g.grd.z);
    } else {
        grdRayHitGrd = lead - g.gro * grdScaledDot;
        groRayHitGrd = g.grd * (-grdScaledDot);
    }
}
// d grayDist -> d grd, d gro: |grd x gro|^2 for the volumetric primitives, |gro + grd ghitT|^2 with ghitT = -gro.z / grd.z for the surfel
__device__ __forceinline__ void gray_distance_grads(bool surfel, const HitGeom& g, float grayGrd, f3& grdGrd, f3& groGrd) {
    if (surfel) {
        const float ghitT = -g.gro.z / g.grd.z;
        const f3 ghitPos = g.gro + g.grd * ghitT;
        const f3 ghitPosGrd = ghitPos * (2.f * grayGrd);
        groGrd = ghitPosGrd;
        grdGrd = ghitPosGrd * ghitT;
        const float ghitTGrd = g.grd.x * ghitPosGrd.x + g.grd.y * ghitPosGrd.y + g.grd.z * ghitPosGrd.z;
        groGrd.z += -ghitTGrd / g.grd.z;
        grdGrd.z += (g.gro.z * ghitTGrd) / (g.grd.z * g.grd.z);
    } else {
        const f3 gcrodGrd = g.gcrod * (2.f * grayGrd);
        grdGrd = mk3(gcrodGrd.z * g.gro.y - gcrodGrd.y * g.gro.z, gcrodGrd.x * g.gro.z - gcrodGrd.z * g.gro.x, gcrodGrd.y * g.gro.x - gcrodGrd.x * g.gro.y);
        groGrd = mk3(gcrodGrd.y * g.grd.z - gcrodGrd.z * g.grd.y, gcrodGrd.z * g.grd.x - gcrodGrd.x * g.grd.z, gcrodGrd.x * g.grd.y - gcrodGrd.y * g.grd.x);
    }
}

// ---------------------------------------------------------------------------------------------
// forward: __raygen__rg of referenceOptix.cu:103-186
// ---------------------------------------------------------------------------------------------
// render.pipeline_type barycentricSurfels (barycentricSurfelsOptix.cu; GrtTraceParams::bary).  A hit of that pipeline is the ray's crossing of
// the surfel's plane - the trisurfel candidate of candidate_abe - and carries the squared distance of the crossing from the surfel's centre in
// the proxy frame (computeTrisurfelSquaredDistance, :179-188, from the triangle's barycentrics: both triangles map onto |(x, y)|^2).  The
// crossing is evaluated again per processed hit, operation by operation as the candidate test (and as the CPU checker's trace_bary_fwd).
__device__ __forceinline__ float surfel_crossing_sqdist(const float* __restrict__ inst, const RayW& r) {
#pragma clang fp contract(off)
    const float4* q = reinterpret_cast<const float4*>(inst);
    const float4 a = q[0], b = q[1], e = q[2];
    const f3 po = proxy_origin(a, b, e, r.o);
    const float pdx = fmaf(a.z, r.d.z, fmaf(a.y, r.d.y, a.x * r.d.x)), pdy = fmaf(b.y, r.d.z, fmaf(b.x, r.d.y, a.w * r.d.x)),
                pdz = fmaf(e.x, r.d.z, fmaf(b.w, r.d.y, b.z * r.d.x));
    const float t = -po.z / pdz;
    const float hx = fmaf(t, pdx, po.x), hy = fmaf(t, pdy, po.y);
    return fmaf(hy, hy, hx * hx);
}
// particleScaledResponse (gaussianParticles.cuh:296-333): the kernel response in the proxy frame, where the unit distance is the particle's
// extent at the density-modulated minimum response
__device__ __forceinline__ float scaled_response(int degree, bool clamped, float gray, float modulated_min_response, float modulation) {
    const float min_response = fminf(modulated_min_response / modulation, 0.97f);
    const float lm = clamped ? logf(min_response) : modulated_min_response;
    switch (degree) {
    case 8: { const float g2 = gray * gray; return expf(lm * g2 * g2); }
    case 5: return expf(lm * gray * gray * sqrtf(gray));
    case 4: return expf(lm * gray * gray);
    case 3: return expf(lm * gray * sqrtf(gray));
    case 1: return expf(lm * sqrtf(gray));
    case 0: { const float sl = (1.f - min_response) / 3.f; return fmaxf(1.f + sl * sqrtf(gray), 0.f); }
    default: return expf(lm * gray);
    }
}

template <int DEG, bool COUNT, bool UNI, bool LOG, bool GEN>
// 4 waves per SIMD (128 VGPRs): the walk is a chain of dependent fetches, a fourth wave hides more of it than the few
// spilled registers cost (measured: 65.8 -> 60.2 ms at 1M particles, 800x800; 5 waves: 69.9 ms; again on the list path at the end of
// round 3: 3 / 4 / 5 waves = 7.46 / 6.85 / 7.47 ms forward)
#ifndef GRT_FWD_WAVES
#define GRT_FWD_WAVES 4
#endif
#ifndef GRT_HIT_PREFETCH
#define GRT_HIT_PREFETCH 0
#endif
// GRT_FWD_PRIO = K > 0: a packet raises its wave's issue priority by one step every K trace rounds (s_setprio 1..3).  A packet's lifetime
// is its number of wave-level tests (correlation 0.99) and nothing known before the launch predicts it (list length: 0.44), but the
// lifetimes are heavy-tailed - a packet that has already run many rounds is one of the long ones - and the launch ends with its longest
// packets (last third of the span at falling occupancy): letting those win the issue arbitration against the three lighter packets of
// their SIMD while the chip is still full shortens the tail without touching the order of the launch (every reordering lost to locality).
#ifndef GRT_FWD_PRIO
#define GRT_FWD_PRIO 0
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GRT_FWD_WAVES, GRT_FWD_WAVES))) void grt_trace_fwd_kernel(GrtTraceParams P, GrtBvh bvh, const float4* __restrict__ density12,
                                                           const float* __restrict__ sph, const float* __restrict__ ray_o,
                                                           const float* __restrict__ ray_d, float* __restrict__ out_rad,
                                                           float* __restrict__ out_dns, float* __restrict__ out_hit2,
                                                           float* __restrict__ out_nrm, float* __restrict__ out_cnt,
                                                           int32_t* __restrict__ visibility, uint32_t* __restrict__ dbg_ids,
                                                           uint32_t* __restrict__ dbg_count, unsigned long long* __restrict__ counters,
                                                           GrtHitLog log, GrtLists lists) {
    __shared__ uint32_t s_stack[UNI ? 1 : kGrtStackDepth];
    __shared__ float s_hit_t[kGrtMaxHits * 64];      // (UNI: the round's staged list entries live here while it is scanned, float4 s_ent[64][3])
    __shared__ uint32_t s_hit_id[kGrtMaxHits * 64];
    static_assert(kGrtMaxHits * 64 * 4 >= 64 * 3 * 16, "the staged list entries must fit the parked hit distances");
    float4* s_ent = reinterpret_cast<float4*>(s_hit_t);
    TraceCounters tc;
    const int lane = threadIdx.x;
    // (starting the heavy packets first was measured slower than this fixed, locality-preserving order although the launch ends with its
    // slowest packet: packets sorted by list length 20.1 ms, dealt alternately from both ends of the ranking 21.4, whole super tiles by
    // total length 20.1, against 18.2-18.6 ms)
    // GEN = false: the instantiation for render.primitive_type instances (the default): every `prim` comparison folds away - the mesh
    // proxies' plane loops, the surfel / trihexa / custom branches of the candidate test and of the per-hit code.  Same arithmetic; the
    // kernel is a fifth shorter and 5 % faster (forward 6.87 -> 6.51 ms at 1 M / 800x800, A/B on one box).  The launcher picks it for the
    // default configuration only: fp32 SH radiance in and out, no normals - those branches fold away with it (the per-ray hit dump of the
    // order tests stays: they must see THIS kernel).
    if (!GEN) { P.prim = GRUT_PRIM_INSTANCES; P.nht = 0; P.sph_half = 0; P.out_half = 0; P.normals = 0; P.bary = 0; }
    // barycentricSurfels: a trace returns the TEN nearest hits (MaxNumHitPerTrace, barycentricSurfelsOptix.cu:26) - the first ten of the
    // sixteen the round gathers; the next round starts behind the tenth
    const int k_round = (GEN && P.bary) ? 10 : kGrtMaxHits;
    const PixelBlock pb = pixel_block(P.W, P.H);
    if (!pb.inside) return;
    const unsigned long long t_begin = COUNT ? wall_clock64() : 0ull;
    const int px = pb.bx * 8 + (lane & 7), py = pb.by * 8 + (lane >> 3);
    const bool in_image = (px < P.W) && (py < P.H);
    const size_t pix = in_image ? (size_t)py * P.W + px : 0;  // out-of-image lanes shadow pixel 0 and never write
    const RayW r = make_ray(P, ray_o, ray_d, pix);
    float list_span = 3.0e38f;   // UNI: how far the previous round reached beyond its start (list_round)
    uint32_t list_end = 0u, list_start = 0u;   // UNI: this packet's candidate list and the scan start (see list_round)
    GrtCone cone = {0.f, 0.f, 1.f, -1.f, 0.f, 1.f, 0.f, 0.f};
    float dmin = 1.f, dmax = 1.f;
    if (UNI) {
        list_start = lists.ranges[2 * (size_t)pb.index];
        list_end = lists.ranges[2 * (size_t)pb.index + 1];
        cone = lists.block_cones[pb.index];
        dmin = __uint_as_float(lists.dir_len_enc[0]); dmax = __uint_as_float(lists.dir_len_enc[1]);
    }
    float basis[16];
    // (filled at the start of each round's processing: 16 registers kept alive across the tree walk cost occupancy)

    f3 rad = mk3(0.f, 0.f, 0.f), nrm = mk3(0.f, 0.f, 0.f);
    float T = 1.f, depth = 0.f, cnt = 0.f;
    float tEnter, tExit;
    scene_interval(bvh.scene, r, tEnter, tExit);
    constexpr float eps = 1e-9f;
    float tLast = fmaxf(0.f, tEnter - eps);
    uint32_t ndbg = 0;
    bool running = in_image;
    const uint32_t block = pb.index;
    uint32_t round = 0;
    bool ghost_overflow = false;   // LOG: a round of this ray saw more ghosts than a chunk holds
    float tmin_prev1 = -3.0e38f, tmin_prev2 = -3.0e38f;   // LOG: tmin of the previous round and of the one before (GhostLog::min_tfar)
    uint32_t last_id = 0xFFFFFFFFu;                       // LOG: the last processed hit is (tLast, last_id)
    uint32_t rounds_done = 0u;                            // GRT_FWD_PRIO

    // one chunk of the hit log per (wave, trace round)
    auto open_chunk = [&]() -> uint32_t* {
        uint32_t c = 0xFFFFFFFFu;
        if (lane == 0) {
            if (round < log.max_rounds) c = atomicAdd(&log.state[0], 1u);
            else log.state[7] = 1u;   // the packet needs more rounds than the table has columns: the host widens the table for the next frame
            if (c >= log.capacity_chunks) { c = 0xFFFFFFFFu; log.state[1] = 1u; }
            if (round < log.max_rounds) log.table[(size_t)block * log.max_rounds + round] = c;
        }
        c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
        round++;
        return c != 0xFFFFFFFFu ? log.pool + (size_t)c * (kGrtLogSlots * 64) + lane : nullptr;
    };

    // The reference traces the 16 nearest candidates beyond the last hit distance, processes them, and traces again
    // (referenceOptix.cu:128-180).
    while (true) {
        running = running && (tLast <= tExit) && (T > P.min_transmittance);
        if (!__any(running)) break;
        // (the round's ghosts are parked in the LDS words of the hit ids, which are only written when the round is over)
        GhostLog ghosts = {s_hit_id + lane, 0u, tmin_prev2, tLast, last_id};
        if (LOG) { tmin_prev2 = tmin_prev1; tmin_prev1 = tLast + eps; }
        uint32_t g_id[kGrtMaxGhosts];
        const unsigned long long ph0 = COUNT ? wall_clock64() : 0ull;
        {
            HitBuffer buf;
            if (UNI) list_round<COUNT, kGrtMaxHits, LOG, true>(lists, cone, dmin, dmax, list_end, list_start, r, tLast + eps, tExit + eps, running, lane, s_ent, buf, tc, &ghosts,
                                                                   s_hit_id + kGrtMaxGhosts * 64, &list_span, P.list_mark);
            else trace_round<COUNT, kGrtMaxHits, LOG>(bvh, r, tLast + eps, tExit + eps, running, lane, s_stack, buf, tc, &ghosts);
            if (LOG) {
#pragma unroll
                for (int g = 0; g < kGrtMaxGhosts; ++g) g_id[g] = (running && (uint32_t)g < ghosts.n) ? s_hit_id[g * 64 + lane] : 0xFFFFFFFFu;
            }
            buf.store(s_hit_t, s_hit_id, lane);
        }
        if (s_hit_id[lane] == 0xFFFFFFFFu) running = false;
        const bool full = s_hit_id[(k_round - 1) * 64 + lane] != 0xFFFFFFFFu;
        const unsigned long long ph1 = COUNT ? wall_clock64() : 0ull;
        // LOG: the round's chunk holds the round's candidates AND its ghosts, merged in (t, particle) order — the replay walks a chunk
        // front to back and decides with the backward program's own intervals which entries that program is offered (the candidates
        // behind the ray's termination are among them: whether the backward reaches one is a matter of ITS interval, t < endT)
        if (LOG) {
            uint32_t* chunk = open_chunk();
            if (ghosts.n > (uint32_t)kGrtMaxGhosts) ghost_overflow = true;
            const float t16 = s_hit_t[(kGrtMaxHits - 1) * 64 + lane];
            const uint32_t id16 = s_hit_id[(kGrtMaxHits - 1) * 64 + lane];
            float g_t[kGrtMaxGhosts];
            uint32_t g_pos[kGrtMaxGhosts];
            uint32_t ng = 0u;
#pragma unroll
            for (int g = 0; g < kGrtMaxGhosts; ++g) {
                const bool have = g_id[g] != 0xFFFFFFFFu;
                g_t[g] = 3.0e38f;
                g_pos[g] = 0u;
                if (!__any(have)) continue;   // (slot g is empty on every ray of the packet - the usual case for the later slots: no distance to evaluate)
                if (have) g_t[g] = candidate(bvh.inst + 12 * (size_t)g_id[g], r, 3.0e38f, 3.0e38f, g_id[g]).t;   // (the proxy index: a trihexa proxy's plane)   // (the very value the round computed: same arithmetic, same inputs)
                // beyond the 16th candidate of a full round: not this round's business (the next round meets it again)
                if (have && full && !hit_less(g_t[g], g_id[g], t16, id16)) { g_id[g] = 0xFFFFFFFFu; g_t[g] = 3.0e38f; }
                ng += g_id[g] != 0xFFFFFFFFu ? 1u : 0u;
            }
            const bool any_ghost = __any(ng != 0u);
            // which ghost slots hold anything on any ray of the packet (wave-uniform): the merge below skips the others - a round's ghosts are
            // few, the slots fill from 0
            uint32_t slot_used = 0u;
#pragma unroll
            for (int g = 0; g < kGrtMaxGhosts; ++g) slot_used |= __any(g_id[g] != 0xFFFFFFFFu) ? (1u << g) : 0u;
            if (chunk) {
#pragma unroll 1
                for (int i = 0; i < kGrtMaxHits; ++i) {
                    const float ft = s_hit_t[i * 64 + lane];
                    const uint32_t fid = s_hit_id[i * 64 + lane];
                    uint32_t pos = (uint32_t)i;   // + the ghosts in front of this candidate
                    if (any_ghost) {
#pragma unroll
                        for (int g = 0; g < kGrtMaxGhosts; ++g) {
                            if (!((slot_used >> g) & 1u)) continue;
                            const bool ghost_first = g_id[g] != 0xFFFFFFFFu && hit_less(g_t[g], g_id[g], ft, fid);
                            pos += ghost_first ? 1u : 0u;
                            g_pos[g] += (g_id[g] != 0xFFFFFFFFu && !ghost_first && fid != 0xFFFFFFFFu) ? 1u : 0u;
                        }
                    }
                    chunk[pos * 64] = running ? fid : 0xFFFFFFFFu;
                }
                if (any_ghost) {
#pragma unroll
                    for (int g = 0; g < kGrtMaxGhosts; ++g) {
                        if (!((slot_used >> g) & 1u)) continue;
#pragma unroll
                        for (int h = 0; h < kGrtMaxGhosts; ++h)
                            if (h != g && ((slot_used >> h) & 1u)) g_pos[g] += (g_id[h] != 0xFFFFFFFFu && hit_less(g_t[h], g_id[h], g_t[g], g_id[g])) ? 1u : 0u;
                        if (g_id[g] != 0xFFFFFFFFu) chunk[g_pos[g] * 64] = g_id[g] | kGrtGhostBit;
                    }
                }
#pragma unroll
                for (int g = 0; g < kGrtMaxGhosts; ++g)
                    if ((uint32_t)g >= ng) chunk[(kGrtMaxHits + g) * 64] = 0xFFFFFFFFu;   // the unused tail of the chunk
            }
        }
        const unsigned long long ph2 = COUNT ? wall_clock64() : 0ull;
        {
            f3 dir = r.d;   // opaque copy: stops the compiler from hoisting the basis back out of the round loop
            asm volatile("" : "+v"(dir.x), "+v"(dir.y), "+v"(dir.z));
            sh_basis16(P.sph_degree, dir, basis);
        }
        // processHit (gaussianParticles.cuh:337-405) for the round's hits in order, while the ray is above min_transmittance
#if GRT_HIT_PREFETCH
        // the next hit's parameter row is requested while this hit is evaluated (a lone wave otherwise waits a memory round trip per hit, and
        // another one for the SH row of an accepted hit: one word of each of its cache lines is touched ahead)
        uint32_t id_next = s_hit_id[lane];
        float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nq = na, ns = na;
        float touch = 0.f;
        auto request = [&](uint32_t idn) {
            if (running && idn != 0xFFFFFFFFu) {
                const uint32_t pn = particle_of(P, idn);
                na = density12[3 * (size_t)pn]; nq = density12[3 * (size_t)pn + 1]; ns = density12[3 * (size_t)pn + 2];
                if (!P.nht && !P.sph_half) { const float* c = sph + (size_t)pn * 3 * P.ncoef; touch = c[0] + c[3 * P.ncoef - 1]; }
            }
        };
        request(id_next);
#endif
#pragma unroll 1
        for (int i = 0; i < k_round; ++i) {
#if GRT_HIT_PREFETCH
            const uint32_t id = id_next;
            const float4 ca_ = na, cq_ = nq, cs_ = ns;
            asm volatile("" :: "v"(touch));
            if (i + 1 < kGrtMaxHits) { id_next = s_hit_id[(i + 1) * 64 + lane]; request(id_next); }
#else
            const uint32_t id = s_hit_id[i * 64 + lane];
#endif
            const float hit_t = s_hit_t[i * 64 + lane];
            const bool process = running && (id != 0xFFFFFFFFu) && (T > P.min_transmittance);
            if (!__any(process)) break;  // ascending list: nothing further for any lane
            if (GEN && P.bary) {
                // barycentricSurfelsOptix.cu:113-166: response from the crossing's squared distance, depth from the HIT distance, the surfel's normal
                if (process) {
                    const float density = density12[3 * (size_t)id].w;
                    const float gray = surfel_crossing_sqdist(bvh.inst + 12 * (size_t)id, r);
                    const bool clamped = P.clamping != 0;
                    const float scale_min = (clamped || DEG == 0) ? P.min_response : logf(P.min_response);
                    const float response = scaled_response(DEG, clamped, gray, scale_min, density);
                    const float alpha = fminf(0.99f, response * density);
                    if ((response > P.min_response) && (alpha > P.min_alpha)) {
                        const float weight = alpha * T;
                        const f3 u = sh_radiance(P, sph, id, basis);
                        rad = rad + mk3(fmaxf(u.x, 0.f), fmaxf(u.y, 0.f), fmaxf(u.z, 0.f)) * weight;
                        T *= (1.f - alpha);
                        depth += hit_t * weight;
                        if (P.normals) {   // the trisurfel kernel's normal: normalize(cross(v1 - v0, v2 - v0)) = -(the particle's third axis)
                            const float4 q = density12[3 * (size_t)id + 1];
                            const m3 rt = quat_wxyz_to_rotT(q.x, q.y, q.z, q.w);
                            const f3 n0 = mk3(-rt.r2.x, -rt.r2.y, -rt.r2.z);
                            nrm = nrm + n0 * ((dot(n0, r.d) < 0.f ? -1.f : 1.f) * weight);
                        }
                        visibility[id] = 1;
                        cnt += 1.f;
                    }
                    tLast = fmaxf(tLast, hit_t);
                    if (COUNT) tc.processed++;
                    if (dbg_ids && ndbg < P.dbg_cap) dbg_ids[pix * P.dbg_cap + ndbg] = id;
                    ndbg++;
                }
                continue;
            }
            if (process) {
                const uint32_t pid = particle_of(P, id);
#if GRT_HIT_PREFETCH
                Particle p;
                p.pos = mk3(ca_.x, ca_.y, ca_.z); p.density = ca_.w; p.quat = cq_; p.scl = mk3(cs_.x, cs_.y, cs_.z);
                p.rotT = quat_wxyz_to_rotT(cq_.x, cq_.y, cq_.z, cq_.w);
#else
                const Particle p = load_particle(density12, pid);
#endif
                const HitGeom g = hit_geometry<DEG>(P, p, r);
                if (g.accept) {
                    const float weight = g.galpha * T;
                    const bool surfel = P.prim == GRUT_PRIM_TRISURFEL;
                    const float pdot = surfel ? -g.gro.z / g.grd.z : -dot(g.grd, g.gro);
                    const f3 grds = p.scl * g.grd * pdot;
                    const float hitT = sqrtf(dot(grds, grds));
                    const f3 u = P.nht ? mk3(0.f, 0.f, 0.f) : sh_radiance(P, sph, pid, basis);   // (nht: the features come from the log, grt_nht_fwd_kernel)
                    const f3 c = mk3(fmaxf(u.x, 0.f), fmaxf(u.y, 0.f), fmaxf(u.z, 0.f));
                    rad = rad + c * weight;
                    T *= (1.f - g.galpha);
                    depth = fmaf(hitT, weight, depth);
                    if (P.normals) {  // gaussianParticles.cuh:398-402
                        const f3 psr = mul_cols(p.rotT, p.scl);
                        if (surfel) {   // the surfel's own normal, towards the side the ray comes from (:398-400)
                            nrm = nrm + mk3(0.f, 0.f, (g.grd.z > 0.f ? 1.f : -1.f) * psr.z) * weight;
                        } else {
                            const f3 q = (g.gro + g.grd * (pdot - sqrtf(9.f - g.gray))) * psr;
                            const float l2 = dot(q, q);
                            const f3 n = l2 > 0.f ? q * (1.f / sqrtf(l2)) : q;
                            nrm = nrm + n * weight;
                        }
                    }
                    visibility[pid] = 1;  // benign race: every writer stores the same value (referenceOptix.cu:158-161)
                    cnt += 1.f;
                }
                tLast = fmaxf(tLast, hit_t);
                if (LOG) last_id = id;
                if (COUNT) tc.processed++;
                if (dbg_ids && ndbg < P.dbg_cap) dbg_ids[pix * P.dbg_cap + ndbg] = particle_of(P, id);
                ndbg++;
            }
        }
        if (!full) running = false;   // the round held every remaining candidate of this ray
        if (COUNT) { const unsigned long long ph3 = wall_clock64(); tc.phase[0] += ph1 - ph0; tc.phase[1] += ph2 - ph1; tc.phase[2] += ph3 - ph2; }
#if GRT_FWD_PRIO > 0
        ++rounds_done;
        if (rounds_done == 1u * GRT_FWD_PRIO) __builtin_amdgcn_s_setprio(1);
        else if (rounds_done == 2u * GRT_FWD_PRIO) __builtin_amdgcn_s_setprio(2);
        else if (rounds_done == 3u * GRT_FWD_PRIO) __builtin_amdgcn_s_setprio(3);
#endif
    }
    if (!in_image) return;
    store_radiance(P, out_rad, pix, rad);
    out_dns[pix] = 1.f - T;
    out_hit2[2 * pix] = depth; out_hit2[2 * pix + 1] = tLast;
    if (P.normals) { out_nrm[3 * pix] = nrm.x; out_nrm[3 * pix + 1] = nrm.y; out_nrm[3 * pix + 2] = nrm.z; }
    if (P.hitcounts) out_cnt[pix] = cnt;
    if (dbg_count) dbg_count[pix] = ndbg;
    if (LOG) log.ray_flags[pix] = ghost_overflow ? kGrtRederiveRay : 0u;   // (a chunk could not hold a round's ghosts: no replay for this ray)
    if (COUNT) {  // work statistics for GrtStats (instrumented launches only)
        if (lane == 0) {   // per block: start and lifetime on the chip-wide 100 MHz counter, node visits (balance analysis, scripts/diag_grt_balance.py)
            const unsigned long long t_end = wall_clock64();
            counters[16 + 3 * (size_t)blockIdx.x] = t_begin;
            counters[16 + 3 * (size_t)blockIdx.x + 1] = t_end - t_begin;
            counters[16 + 3 * (size_t)blockIdx.x + 2] = ((unsigned long long)tc.nodes << 32) | tc.wave_leaves;
        }
        atomicAdd(&counters[0], UNI ? (lane == 0 ? tc.test_ticks : 0ull) : (unsigned long long)tc.nodes);   // (packet lists visit no node: the word carries the work loops' ticks)
        atomicAdd(&counters[1], (unsigned long long)tc.leaf_tests);
        atomicAdd(&counters[2], (unsigned long long)tc.processed);
        atomicAdd(&counters[3], (unsigned long long)tc.rounds);
        atomicAdd(&counters[4], (unsigned long long)tc.inserts);
        for (int k = 0; k < 4; ++k) atomicAdd(&counters[5 + k], (unsigned long long)tc.rej[k]);   // [5] = passed all three tests
        atomicAdd(&counters[9], (unsigned long long)tc.wave_leaves);
        atomicAdd(&counters[10], (unsigned long long)tc.wave_slab);
        atomicAdd(&counters[11], (unsigned long long)tc.wave_insert);
        atomicAdd(&counters[12], (unsigned long long)tc.batch_loads);
        if (lane == 0) for (int k = 0; k < 3; ++k) atomicAdd(&counters[13 + k], tc.phase[k]);
    }
}

// ---------------------------------------------------------------------------------------------
// backward: __raygen__rg of referenceBwdOptix.cu:103-170 + processHitBwd (gaussianParticles.cuh:468-731)
// ---------------------------------------------------------------------------------------------
// gradient of (p * rotT(q)) . g w.r.t. q (matmul_bw_quat, mathUtils.h)
__device__ __forceinline__ float4 matmul_bw_quat(f3 p, f3 g, float4 q) {
    const f3 d0 = p * g.x, d1 = p * g.y, d2 = p * g.z;
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    float dr = 0.f, dx = 0.f, dy = 0.f, dz = 0.f;
    dy += -4.f * y * d0.x; dz += -4.f * z * d0.x;
    dr += 2.f * z * d0.y; dx += 2.f * y * d0.y; dy += 2.f * x * d0.y; dz += 2.f * r * d0.y;
    dr += -2.f * y * d0.z; dx += 2.f * z * d0.z; dy += -2.f * r * d0.z; dz += 2.f * x * d0.z;
    dr += -2.f * z * d1.x; dx += 2.f * y * d1.x; dy += 2.f * x * d1.x; dz += -2.f * r * d1.x;
    dx += -4.f * x * d1.y; dz += -4.f * z * d1.y;
    dr += 2.f * x * d1.z; dx += 2.f * r * d1.z; dy += 2.f * z * d1.z; dz += 2.f * y * d1.z;
    dr += 2.f * y * d2.x; dx += 2.f * z * d2.x; dy += 2.f * r * d2.x; dz += 2.f * x * d2.x;
    dr += -2.f * x * d2.y; dx += -2.f * r * d2.y; dy += 2.f * z * d2.y; dz += 2.f * y * d2.y;
    dx += -4.f * x * d2.z; dy += -4.f * y * d2.z;
    return make_float4(dr, dx, dy, dz);
}

struct BwdRay {
    f3 rad_fin, rad_grad;
    float T_fin, depth_fin, T_grad, depth_grad;
    f3 rad;          // running
    float T, depth;  // running
};
// processHitBwd (gaussianParticles.cuh:468-731) for one ray and one particle, gradients added with atomics
template <int DEG>
__device__ __forceinline__ void process_hit_bwd(const GrtTraceParams& P, const RayW& r, const float basis[16], int nact, uint32_t id,
                                                const float4* __restrict__ density12, const float* __restrict__ sph, BwdRay& st,
                                                float* __restrict__ g_density12, float* __restrict__ g_sph) {
    const f3 rad_fin = st.rad_fin, rad_grad = st.rad_grad;
    const float T_fin = st.T_fin, depth_fin = st.depth_fin, T_grad = st.T_grad, depth_grad = st.depth_grad;
    f3 rad = st.rad;
    float T = st.T, depth = st.depth;
    const Particle p = load_particle(density12, id);
    const HitGeom g = hit_geometry<DEG>(P, p, r);
    if (g.accept) {
        const f3 gscl = p.scl;
        const bool surfel = P.prim == GRUT_PRIM_TRISURFEL;
        const float pdot = surfel ? -g.gro.z / g.grd.z : -dot(g.grd, g.gro);
        const f3 grdd = g.grd * pdot;
        const f3 grds = gscl * grdd;
        const float gsq = dot(grds, grds);
        const float gdist = sqrtf(gsq);
        const float weight = g.galpha * T;
        const float nextT = (1.f - g.galpha) * T;
        depth = fmaf(weight, gdist, depth);
        const float resHitT = fmaxf(nextT <= P.min_transmittance ? 0.f : (depth_fin - depth) / nextT, 0.f);
        const float galphaRayHitGrd = (gdist - resHitT) * T * depth_grad;
        const f3 grdsRayHitGrd = gsq > 0.f ? grds * ((2.f * weight) / (2.f * gdist) * depth_grad) : mk3(0.f, 0.f, 0.f);
        const f3 gsclRayHitGrd = grdd * grdsRayHitGrd;
        const float grdScaledDot = dot(grdsRayHitGrd * gscl, g.grd);
        f3 grdRayHitGrd, groRayHitGrd;
        surfel_hit_distance_grads(surfel, g, gscl * grdsRayHitGrd * pdot, pdot, grdScaledDot, grdRayHitGrd, groRayHitGrd);
        const float resTrm = g.galpha < 0.999999f ? T_fin / (1.f - g.galpha) : T;
        const float galphaRayDnsGrd = resTrm * -T_grad;

        // radianceFromSpHBwd (:101-177)
        const f3 gradu = sh_radiance(P, sph, id, basis);
        const f3 grad = mk3(fmaxf(gradu.x, 0.f), fmaxf(gradu.y, 0.f), fmaxf(gradu.z, 0.f));
        f3 dL = rad_grad * weight;
        if (!(gradu.x > 0.f)) dL.x = 0.f;
        if (!(gradu.y > 0.f)) dL.y = 0.f;
        if (!(gradu.z > 0.f)) dL.z = 0.f;
        float* gs = g_sph + (size_t)id * 3 * P.ncoef;
        for (int k = 0; k < nact; ++k) {
            atomicAdd(gs + 3 * k, basis[k] * dL.x);
            atomicAdd(gs + 3 * k + 1, basis[k] * dL.y);
            atomicAdd(gs + 3 * k + 2, basis[k] * dL.z);
        }
        rad = rad + grad * weight;
        f3 resRad = mk3(0.f, 0.f, 0.f);
        if (!(nextT <= P.min_transmittance)) {
            const float inT = 1.f / nextT;
            resRad = mk3(fmaxf((rad_fin.x - rad.x) * inT, 0.f), fmaxf((rad_fin.y - rad.y) * inT, 0.f), fmaxf((rad_fin.z - rad.z) * inT, 0.f));
        }
        const float common = galphaRayHitGrd + galphaRayDnsGrd + T * (grad.x - resRad.x) * rad_grad.x + T * (grad.y - resRad.y) * rad_grad.y +
                             T * (grad.z - resRad.z) * rad_grad.z;
        float* gd = g_density12 + 12 * (size_t)id;
        atomicAdd(gd + 3, g.gres * common);
        const float gresGrd = p.density * common;
        const float grayGrd = particle_response_grd<DEG>(g.gray, g.gres, gresGrd);
        f3 grdGrd, groGrd;
        gray_distance_grads(surfel, g, grayGrd, grdGrd, groGrd);
        const f3 groTot = groGrd + groRayHitGrd;
        const f3 is2 = g.giscl * g.giscl;
        const f3 gsclGrdGro = mk3(-g.gposcr.x * is2.x, -g.gposcr.y * is2.y, -g.gposcr.z * is2.z) * groTot;
        const f3 gposcrGrd = g.giscl * groTot;
        const f3 gposcGrd = mul_cols(p.rotT, gposcrGrd);
        const float4 grotGrdPoscr = matmul_bw_quat(g.gposc, gposcrGrd, p.quat);
        atomicAdd(gd + 0, -gposcGrd.x); atomicAdd(gd + 1, -gposcGrd.y); atomicAdd(gd + 2, -gposcGrd.z);
        // safe_normalize_bw
        const f3 dn = grdGrd + grdRayHitGrd;
        const float l2 = dot(g.grdu, g.grdu);
        f3 grduGrd = mk3(0.f, 0.f, 0.f);
        if (l2 > 0.f) {
            const float il = 1.f / sqrtf(l2), il3 = il * il * il;
            const float sdot = dot(dn, g.grdu);
            grduGrd = dn * il - g.grdu * (il3 * sdot);
        }
        atomicAdd(gd + 8, gsclRayHitGrd.x + gsclGrdGro.x + (-g.rdr.x * is2.x) * grduGrd.x);
        atomicAdd(gd + 9, gsclRayHitGrd.y + gsclGrdGro.y + (-g.rdr.y * is2.y) * grduGrd.y);
        atomicAdd(gd + 10, gsclRayHitGrd.z + gsclGrdGro.z + (-g.rdr.z * is2.z) * grduGrd.z);
        const f3 rdrGrd = g.giscl * grduGrd;
        const float4 grotGrdRd = matmul_bw_quat(r.d, rdrGrd, p.quat);
        atomicAdd(gd + 4, grotGrdPoscr.x + grotGrdRd.x); atomicAdd(gd + 5, grotGrdPoscr.y + grotGrdRd.y);
        atomicAdd(gd + 6, grotGrdPoscr.z + grotGrdRd.z); atomicAdd(gd + 7, grotGrdPoscr.w + grotGrdRd.w);
        T = nextT;
    }
    st.rad = rad; st.T = T; st.depth = depth;
}

// ---------------------------------------------------------------------------------------------
// Neural harmonic features (model.feature_type = nht, the Slang pipelines: referenceSlangOptix.cu:103-186): the per-ray features are not a
// per-particle colour but are interpolated per hit at the hit's canonical intersection (neuralHarmonicFeaturesParticle.slang:146-196,
// see gut_render.hip: gut_render_nht_fwd_kernel for the model).  The trace kernel stays what it is — transmittance, hit distance,
// counts, visibility, and the LOG of every round's candidates — and this pass walks each ray's log like the forward did (the round's
// candidates in hit order, processed while the ray is above min_transmittance; the same arithmetic, so the same decisions) and
// integrates ray_dim features per ray: the trace kernel's register budget (128, four waves per SIMD) has no room for 24 accumulators.
// ---------------------------------------------------------------------------------------------
constexpr int kGrtNhtMaxRay = 32, kGrtNhtMaxIpd = 16;
__device__ __forceinline__ float grt_nht_sin(float x) { return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x * 0.15915494309189535f)); }
__device__ __forceinline__ float grt_nht_cos(float x) { return __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(x * 0.15915494309189535f)); }
struct NhtTetra {
    f3 v0, e1, e2, e3, c23, gw0, gw1, gw2, gw3;
    float inv_det;
};
__device__ __forceinline__ NhtTetra nht_tetra() {
    NhtTetra t;
    const float edge = 4.898979485566356f, face_h = 4.242640687119285f, face_in = 1.4142135623730951f;
    t.v0 = mk3(0.5f * edge, -face_in, -1.f);
    const f3 v1 = mk3(-0.5f * edge, -face_in, -1.f), v2 = mk3(0.f, face_h - face_in, -1.f), v3 = mk3(0.f, 0.f, 3.f);
    t.e1 = v1 - t.v0; t.e2 = v2 - t.v0; t.e3 = v3 - t.v0;
    t.c23 = cross(t.e2, t.e3);
    t.inv_det = 1.f / dot(t.e1, t.c23);
    t.gw1 = t.c23 * t.inv_det; t.gw2 = cross(t.e3, t.e1) * t.inv_det; t.gw3 = cross(t.e1, t.e2) * t.inv_det;
    t.gw0 = (t.gw1 + t.gw2 + t.gw3) * -1.f;
    return t;
}
__device__ __forceinline__ void nht_weights(const GrtTraceParams& P, const NhtTetra& t, f3 Pc, float (&wq)[4]) {
    wq[0] = 1.f; wq[1] = wq[2] = wq[3] = 0.f;
    if (P.nht_support == 1) {
        const f3 d = Pc - t.v0;
        wq[1] = dot(d, t.c23) * t.inv_det; wq[2] = dot(t.e1, cross(d, t.e3)) * t.inv_det; wq[3] = dot(t.e1, cross(t.e2, d)) * t.inv_det;
        wq[0] = 1.f - wq[1] - wq[2] - wq[3];
    }
}
__device__ __forceinline__ float nht_feature_value(const GrtTraceParams& P, const float* __restrict__ features, uint32_t id, int word) {
    const size_t at = (size_t)id * P.nht_k + word;
    return P.sph_half ? __half2float(reinterpret_cast<const __half*>(features)[at]) : features[at];
}
__device__ __forceinline__ void nht_base(const GrtTraceParams& P, const float* __restrict__ features, uint32_t id, const float (&wq)[4], float (&base)[kGrtNhtMaxIpd]) {
    const int points = P.nht_support == 1 ? 4 : 1;
#pragma unroll
    for (int m = 0; m < kGrtNhtMaxIpd; ++m) {
        base[m] = 0.f;
        if (m < P.nht_ipd)
            for (int k = 0; k < points; ++k) {
                const float fv = nht_feature_value(P, features, id, k * P.nht_ipd + m);
                base[m] = k == 0 ? fv * wq[0] : fmaf(wq[k], fv, base[m]);
            }
    }
}
// feature i of the activated vector and its derivative w.r.t. its base feature kb (sincos: i = kb*nf*2 + f*2 + {0,1}; siren: i = kb*nf + f)
__device__ __forceinline__ void nht_activation(const GrtTraceParams& P, const float (&base)[kGrtNhtMaxIpd], int i, float& f, float& df, int& kb) {
    const int nf = P.nht_nf;
    if (P.nht_act == 0) { kb = i; f = base[kb < kGrtNhtMaxIpd ? kb : 0]; df = 1.f; }
    else if (P.nht_act == 3) { kb = i; const float bv = base[kb < kGrtNhtMaxIpd ? kb : 0]; f = fmaxf(0.f, bv); df = bv > 0.f ? 1.f : 0.f; }
    else if (P.nht_act == 2) {
        kb = i / (2 * nf);
        const int rem = i - kb * 2 * nf, fq = rem >> 1;
        const float fr = (float)(fq + 1), ang = base[kb < kGrtNhtMaxIpd ? kb : 0] * fr;
        const float sn = grt_nht_sin(ang), cs = grt_nht_cos(ang);
        f = (rem & 1) ? cs : sn; df = (rem & 1) ? -fr * sn : fr * cs;
    } else {
        kb = i / nf;
        const float fr = ldexpf(1.f, i - kb * nf), ang = base[kb < kGrtNhtMaxIpd ? kb : 0] * fr;
        f = grt_nht_sin(ang); df = fr * grt_nht_cos(ang);
    }
}
// ---- the default feature model with compile-time shapes (configs/base_gs.yaml:96-103: 48 = 4 vertices x 12 floats, sincos, one frequency
// -> 24 ray features): the particle's row is fetched ONCE as twelve 16-byte loads and everything else stays in registers.  The generic
// functions above index the row word by word from memory inside run-time loops (96 scattered loads and a 32 x 16 select cascade per hit).
constexpr int kGrtNhtFastIpd = 12, kGrtNhtFastRay = 24, kGrtNhtFastK = 48;
__device__ __forceinline__ bool grt_nht_fast_shape(const GrtTraceParams& P) {
    return P.nht_k == kGrtNhtFastK && P.nht_ipd == kGrtNhtFastIpd && P.nht_support == 1 && P.nht_act == 2 && P.nht_nf == 1;
}
__device__ __forceinline__ void nht_load_row48(const GrtTraceParams& P, const float* __restrict__ features, uint32_t id, float (&F)[kGrtNhtFastK]) {
    if (P.sph_half) {
        const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const __half*>(features) + (size_t)id * kGrtNhtFastK);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const uint4 w = src[q];
            const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const __half2 h2 = *reinterpret_cast<const __half2*>(&ws[k]);
                F[8 * q + 2 * k] = __low2float(h2);
                F[8 * q + 2 * k + 1] = __high2float(h2);
            }
        }
    } else {
        const float4* src = reinterpret_cast<const float4*>(features + (size_t)id * kGrtNhtFastK);
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const float4 w = src[q];
            F[4 * q] = w.x; F[4 * q + 1] = w.y; F[4 * q + 2] = w.z; F[4 * q + 3] = w.w;
        }
    }
}
__device__ __forceinline__ void nht_fast_sincos(const float (&F)[kGrtNhtFastK], const float (&wq)[4], float (&sn)[kGrtNhtFastIpd], float (&cs)[kGrtNhtFastIpd]) {
#pragma unroll
    for (int m = 0; m < kGrtNhtFastIpd; ++m) {
        float base = F[m] * wq[0];
        base = fmaf(wq[1], F[kGrtNhtFastIpd + m], base);
        base = fmaf(wq[2], F[2 * kGrtNhtFastIpd + m], base);
        base = fmaf(wq[3], F[3 * kGrtNhtFastIpd + m], base);
        sn[m] = grt_nht_sin(base);
        cs[m] = grt_nht_cos(base);
    }
}

template <int DEG, bool FAST = false>
__global__ __launch_bounds__(64) void grt_nht_fwd_kernel(GrtTraceParams P, const float4* __restrict__ density12, const float* __restrict__ features,
                                                         const float* __restrict__ ray_o, const float* __restrict__ ray_d, float* __restrict__ out_feat,
                                                         GrtHitLog log) {
    const int lane = threadIdx.x;
    const PixelBlock pb = pixel_block(P.W, P.H);
    if (!pb.inside) return;
    const int px = pb.bx * 8 + (lane & 7), py = pb.by * 8 + (lane >> 3);
    const bool in_image = (px < P.W) && (py < P.H);
    const size_t pix = in_image ? (size_t)py * P.W + px : 0;
    const RayW r = make_ray(P, ray_o, ray_d, pix);
    const NhtTetra tet = nht_tetra();
    const int nr = P.nht_ray_dim;
    float acc[kGrtNhtMaxRay];
#pragma unroll
    for (int i = 0; i < kGrtNhtMaxRay; ++i) acc[i] = 0.f;
    float T = 1.f;
    for (uint32_t round = 0; round < log.max_rounds; ++round) {
        const uint32_t c = log.table[(size_t)pb.index * log.max_rounds + round];
        if (c == 0xFFFFFFFFu) break;
        const uint32_t* chunk = log.pool + (size_t)c * (kGrtLogSlots * 64) + lane;
#pragma unroll 1
        for (int i = 0; i < kGrtLogSlots; ++i) {
            const uint32_t slot = in_image ? chunk[i * 64] : 0xFFFFFFFFu;
            if (!__any(slot != 0xFFFFFFFFu)) break;
            // the forward processed the round's CANDIDATES (not its ghosts) in this order while the ray was above min_transmittance
            if (slot != 0xFFFFFFFFu && !(slot & kGrtGhostBit) && (T > P.min_transmittance)) {
                const uint32_t id = particle_of(P, slot);   // (the log is keyed by proxy: trihexa / sphere hold several per particle)
                const Particle p = load_particle(density12, id);
                const HitGeom g = hit_geometry<DEG>(P, p, r);
                if (g.accept) {
                    const float weight = g.galpha * T;
                    T *= (1.f - g.galpha);
                    if (FAST && weight > 0.f) {
                        const float pdot = -dot(g.grd, g.gro);
                        float wq[4], F[kGrtNhtFastK], sn[kGrtNhtFastIpd], cs[kGrtNhtFastIpd];
                        nht_weights(P, tet, g.gro + g.grd * pdot, wq);
                        nht_load_row48(P, features, id, F);
                        nht_fast_sincos(F, wq, sn, cs);
#pragma unroll
                        for (int m = 0; m < kGrtNhtFastIpd; ++m) {
                            acc[2 * m] = fmaf(sn[m], weight, acc[2 * m]);
                            acc[2 * m + 1] = fmaf(cs[m], weight, acc[2 * m + 1]);
                        }
                    } else if (weight > 0.f) {
                        const float pdot = -dot(g.grd, g.gro);
                        float wq[4], base[kGrtNhtMaxIpd];
                        nht_weights(P, tet, g.gro + g.grd * pdot, wq);
                        nht_base(P, features, id, wq, base);
#pragma unroll
                        for (int k = 0; k < kGrtNhtMaxRay; ++k)
                            if (k < nr) {
                                float f, df;
                                int kb;
                                nht_activation(P, base, k, f, df, kb);
                                acc[k] = fmaf(f, weight, acc[k]);
                            }
                    }
                }
            }
        }
    }
    if (!in_image) return;
#pragma unroll
    for (int k = 0; k < kGrtNhtMaxRay; ++k)
        if (k < nr) {
            if (P.out_half) reinterpret_cast<__half*>(out_feat)[pix * nr + k] = __float2half(acc[k]);
            else out_feat[pix * nr + k] = acc[k];
        }
}
// One hit of the Slang backward pipeline with neural harmonic features (referenceSlangBwdOptix.cu:118-178): particleDensityHit,
// particleFeaturesFromBuffer, particleFeaturesIntegrateBwdToBuffer (the lerp form un-blended front to back) and
// particleDensityProcessHitBwdToBuffer with the canonical intersection's gradient — the reverse mode restated by the CPU checker
// (orc_grt_trace_nht_bwd; same formulas as the 3DGUT nht backward, which float64 autograd pins).  Advances the ray state and returns the
// hit's gradient as {11 geometric terms, the four barycentric weights, d L / d base feature}: the feature rows' gradient is
// wq[k] * gbase[n] for row word k * ipd + n.
struct NhtBwdRay {
    float Cb[kGrtNhtMaxRay], gC[kGrtNhtMaxRay];   // "behind" features (start at the forward's result) and their running upstream gradient
    float Tb, gT, Db, gD;
    // fast shape (nht_hit_bwd_fast): every gC_i is scaled by the same (1 - alpha) per hit, so gC stays the UPSTREAM gradient and G carries
    // the common factor; the 24 "behind" features enter only through S = sum_i Cb_i gC_i, un-blended as one scalar
    float S, G;
};
template <int DEG>
__device__ __forceinline__ bool nht_hit_bwd(const GrtTraceParams& P, const RayW& r, const Particle& p, const HitGeom& g, const float* __restrict__ features,
                                            uint32_t id, const NhtTetra& tet, NhtBwdRay& st, float (&gd)[11], float (&wq)[4], float (&gbase)[kGrtNhtMaxIpd]) {
#pragma unroll
    for (int m = 0; m < kGrtNhtMaxIpd; ++m) gbase[m] = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) gd[k] = 0.f;
    wq[0] = 1.f; wq[1] = wq[2] = wq[3] = 0.f;
    if (!g.accept) return false;
    const float alpha = g.galpha;
    const float pdot = -dot(g.grd, g.gro);
    const f3 grdd = g.grd * pdot;
    const f3 grds = p.scl * grdd;
    const float gsq = dot(grds, grds);
    const float hitT = sqrtf(gsq);
    nht_weights(P, tet, g.gro + grdd, wq);
    float base[kGrtNhtMaxIpd];
    nht_base(P, features, id, wq, base);
    const float w = 1.f / (1.f - alpha);
    const bool contributes = alpha > 0.f;
    float dalpha = 0.f;
#pragma unroll
    for (int i = 0; i < kGrtNhtMaxRay; ++i)
        if (i < P.nht_ray_dim && contributes) {
            float f, df;
            int kb;
            nht_activation(P, base, i, f, df, kb);
            st.Cb[i] = (st.Cb[i] - f * alpha) * w;
            dalpha = fmaf(f - st.Cb[i], st.gC[i], dalpha);
            const float gf = alpha * st.gC[i];
            st.gC[i] *= (1.f - alpha);
#pragma unroll
            for (int m = 0; m < kGrtNhtMaxIpd; ++m)
                if (m == kb) gbase[m] = fmaf(df, gf, gbase[m]);
        }
    f3 dP = mk3(0.f, 0.f, 0.f);
    if (P.nht_support == 1) {
        float dw[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < kGrtNhtMaxIpd; ++m)
            if (m < P.nht_ipd)
                for (int k = 0; k < 4; ++k) dw[k] = fmaf(nht_feature_value(P, features, id, k * P.nht_ipd + m), gbase[m], dw[k]);
        dP = tet.gw0 * dw[0] + tet.gw1 * dw[1] + tet.gw2 * dw[2] + tet.gw3 * dw[3];
    }
    st.Tb *= w;
    st.Db = (st.Db - hitT * alpha) * w;
    dalpha += (hitT - st.Db) * st.gD - st.Tb * st.gT;
    const float ddepth = alpha * st.gD;
    st.gD *= (1.f - alpha);
    st.gT *= (1.f - alpha);
    float dres = 0.f, ddens = 0.f;
    if (g.gres * p.density < P.max_alpha) { dres = p.density * dalpha; ddens = g.gres * dalpha; }
    const float grayGrd = particle_response_grd<DEG>(g.gray, g.gres, dres);
    const f3 grdsGrd = gsq > 0.f ? grds * (ddepth / hitT) : mk3(0.f, 0.f, 0.f);
    const f3 gsclHit = grdd * grdsGrd;
    const float sdot = dot(grdsGrd * p.scl, g.grd);
    const float gdP = dot(g.grd, dP);
    const f3 grdHit = p.scl * grdsGrd * pdot - g.gro * sdot + dP * pdot - g.gro * gdP;
    const f3 groHit = g.grd * (-sdot) + dP - g.grd * gdP;
    const f3 gcrodGrd = g.gcrod * (2.f * grayGrd);
    const f3 grdGrd = mk3(gcrodGrd.z * g.gro.y - gcrodGrd.y * g.gro.z, gcrodGrd.x * g.gro.z - gcrodGrd.z * g.gro.x, gcrodGrd.y * g.gro.x - gcrodGrd.x * g.gro.y);
    const f3 groGrd = mk3(gcrodGrd.y * g.grd.z - gcrodGrd.z * g.grd.y, gcrodGrd.z * g.grd.x - gcrodGrd.x * g.grd.z, gcrodGrd.x * g.grd.y - gcrodGrd.y * g.grd.x);
    const f3 groTot = groGrd + groHit;
    const f3 is2 = g.giscl * g.giscl;
    const f3 gsclGro = mk3(-g.gposcr.x * is2.x, -g.gposcr.y * is2.y, -g.gposcr.z * is2.z) * groTot;
    const f3 gposcrGrd = g.giscl * groTot;
    const f3 gposcGrd = mul_cols(p.rotT, gposcrGrd);
    const float4 gq1 = matmul_bw_quat(g.gposc, gposcrGrd, p.quat);
    const f3 dn = grdGrd + grdHit;
    const float l2 = dot(g.grdu, g.grdu);
    f3 grduGrd = mk3(0.f, 0.f, 0.f);
    if (l2 > 0.f) {
        const float il = 1.f / sqrtf(l2);
        grduGrd = dn * il - g.grdu * (il * il * il * dot(dn, g.grdu));
    }
    const f3 sclGrd = gsclHit + gsclGro + mk3(-g.rdr.x * is2.x, -g.rdr.y * is2.y, -g.rdr.z * is2.z) * grduGrd;
    const float4 gq2 = matmul_bw_quat(r.d, g.giscl * grduGrd, p.quat);
    gd[0] = -gposcGrd.x; gd[1] = -gposcGrd.y; gd[2] = -gposcGrd.z; gd[3] = ddens;
    gd[4] = gq1.x + gq2.x; gd[5] = gq1.y + gq2.y; gd[6] = gq1.z + gq2.z; gd[7] = gq1.w + gq2.w;
    gd[8] = sclGrd.x; gd[9] = sclGrd.y; gd[10] = sclGrd.z;
    return contributes;
}
template <int DEG>
__device__ __forceinline__ bool nht_hit_bwd_fast(const GrtTraceParams& P, const RayW& r, const Particle& p, const HitGeom& g, const float* __restrict__ features,
                                                 uint32_t id, const NhtTetra& tet, NhtBwdRay& st, float (&gd)[11], float (&wq)[4], float (&gbase)[kGrtNhtMaxIpd]) {
#pragma unroll
    for (int m = 0; m < kGrtNhtMaxIpd; ++m) gbase[m] = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) gd[k] = 0.f;
    wq[0] = 1.f; wq[1] = wq[2] = wq[3] = 0.f;
    if (!g.accept) return false;
    const float alpha = g.galpha;
    const float pdot = -dot(g.grd, g.gro);
    const f3 grdd = g.grd * pdot;
    const f3 grds = p.scl * grdd;
    const float gsq = dot(grds, grds);
    const float hitT = sqrtf(gsq);
    nht_weights(P, tet, g.gro + grdd, wq);
    const float w = 1.f / (1.f - alpha);
    const bool contributes = alpha > 0.f;
    float dalpha = 0.f;
    f3 dP = mk3(0.f, 0.f, 0.f);
    if (contributes) {
        float F[kGrtNhtFastK], sn[kGrtNhtFastIpd], cs[kGrtNhtFastIpd];
        nht_load_row48(P, features, id, F);
        nht_fast_sincos(F, wq, sn, cs);
        float fg = 0.f;
#pragma unroll
        for (int m = 0; m < kGrtNhtFastIpd; ++m) fg = fmaf(sn[m], st.gC[2 * m], fmaf(cs[m], st.gC[2 * m + 1], fg));
        st.S = (st.S - alpha * fg) * w;
        dalpha = st.G * (fg - st.S);
        const float ag = alpha * st.G;
        float dw[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < kGrtNhtFastIpd; ++m) {
            gbase[m] = ag * (cs[m] * st.gC[2 * m] - sn[m] * st.gC[2 * m + 1]);
            dw[0] = fmaf(F[m], gbase[m], dw[0]); dw[1] = fmaf(F[kGrtNhtFastIpd + m], gbase[m], dw[1]);
            dw[2] = fmaf(F[2 * kGrtNhtFastIpd + m], gbase[m], dw[2]); dw[3] = fmaf(F[3 * kGrtNhtFastIpd + m], gbase[m], dw[3]);
        }
        st.G *= (1.f - alpha);
        dP = tet.gw0 * dw[0] + tet.gw1 * dw[1] + tet.gw2 * dw[2] + tet.gw3 * dw[3];
    }
    st.Tb *= w;
    st.Db = (st.Db - hitT * alpha) * w;
    dalpha += (hitT - st.Db) * st.gD - st.Tb * st.gT;
    const float ddepth = alpha * st.gD;
    st.gD *= (1.f - alpha);
    st.gT *= (1.f - alpha);
    float dres = 0.f, ddens = 0.f;
    if (g.gres * p.density < P.max_alpha) { dres = p.density * dalpha; ddens = g.gres * dalpha; }
    const float grayGrd = particle_response_grd<DEG>(g.gray, g.gres, dres);
    const f3 grdsGrd = gsq > 0.f ? grds * (ddepth / hitT) : mk3(0.f, 0.f, 0.f);
    const f3 gsclHit = grdd * grdsGrd;
    const float sdot = dot(grdsGrd * p.scl, g.grd);
    const float gdP = dot(g.grd, dP);
    const f3 grdHit = p.scl * grdsGrd * pdot - g.gro * sdot + dP * pdot - g.gro * gdP;
    const f3 groHit = g.grd * (-sdot) + dP - g.grd * gdP;
    const f3 gcrodGrd = g.gcrod * (2.f * grayGrd);
    const f3 grdGrd = mk3(gcrodGrd.z * g.gro.y - gcrodGrd.y * g.gro.z, gcrodGrd.x * g.gro.z - gcrodGrd.z * g.gro.x, gcrodGrd.y * g.gro.x - gcrodGrd.x * g.gro.y);
    const f3 groGrd = mk3(gcrodGrd.y * g.grd.z - gcrodGrd.z * g.grd.y, gcrodGrd.z * g.grd.x - gcrodGrd.x * g.grd.z, gcrodGrd.x * g.grd.y - gcrodGrd.y * g.grd.x);
    const f3 groTot = groGrd + groHit;
    const f3 is2 = g.giscl * g.giscl;
    const f3 gsclGro = mk3(-g.gposcr.x * is2.x, -g.gposcr.y * is2.y, -g.gposcr.z * is2.z) * groTot;
    const f3 gposcrGrd = g.giscl * groTot;
    const f3 gposcGrd = mul_cols(p.rotT, gposcrGrd);
    const float4 gq1 = matmul_bw_quat(g.gposc, gposcrGrd, p.quat);
    const f3 dn = grdGrd + grdHit;
    const float l2 = dot(g.grdu, g.grdu);
    f3 grduGrd = mk3(0.f, 0.f, 0.f);
    if (l2 > 0.f) {
        const float il = 1.f / sqrtf(l2);
        grduGrd = dn * il - g.grdu * (il * il * il * dot(dn, g.grdu));
    }
    const f3 sclGrd = gsclHit + gsclGro + mk3(-g.rdr.x * is2.x, -g.rdr.y * is2.y, -g.rdr.z * is2.z) * grduGrd;
    const float4 gq2 = matmul_bw_quat(r.d, g.giscl * grduGrd, p.quat);
    gd[0] = -gposcGrd.x; gd[1] = -gposcGrd.y; gd[2] = -gposcGrd.z; gd[3] = ddens;
    gd[4] = gq1.x + gq2.x; gd[5] = gq1.y + gq2.y; gd[6] = gq1.z + gq2.z; gd[7] = gq1.w + gq2.w;
    gd[8] = sclGrd.x; gd[9] = sclGrd.y; gd[10] = sclGrd.z;
    return contributes;
}
__device__ __forceinline__ NhtBwdRay nht_bwd_ray_init(const GrtTraceParams& P, bool use, size_t pix, const float* __restrict__ in_feat,
                                                      const float* __restrict__ in_dns, const float* __restrict__ in_hit2,
                                                      const float* __restrict__ g_feat, const float* __restrict__ g_dns, const float* __restrict__ g_hit) {
    NhtBwdRay st;
    const int nr = P.nht_ray_dim;
#pragma unroll
    for (int i = 0; i < kGrtNhtMaxRay; ++i) {
        const bool u = use && i < nr;
        st.Cb[i] = u ? (P.out_half ? __half2float(reinterpret_cast<const __half*>(in_feat)[pix * nr + i]) : in_feat[pix * nr + i]) : 0.f;
        st.gC[i] = u ? g_feat[pix * nr + i] : 0.f;
    }
    st.Tb = 1.f - in_dns[pix]; st.gT = -g_dns[pix];
    st.Db = in_hit2[2 * pix]; st.gD = g_hit ? g_hit[pix] : 0.f;
    st.S = 0.f; st.G = 1.f;
#pragma unroll
    for (int i = 0; i < kGrtNhtMaxRay; ++i) st.S = fmaf(st.Cb[i], st.gC[i], st.S);
    return st;
}
// lane-level version for the traversal backward (the rays the replay does not serve): the reference's per-hit atomics
template <int DEG>
__device__ __forceinline__ void process_hit_bwd_nht(const GrtTraceParams& P, const RayW& r, uint32_t id, const float4* __restrict__ density12,
                                                    const float* __restrict__ features, const NhtTetra& tet, NhtBwdRay& st,
                                                    float* __restrict__ g_density12, float* __restrict__ g_features) {
    id = particle_of(P, id);   // (the traversal hands over the proxy)
    const Particle p = load_particle(density12, id);
    const HitGeom g = hit_geometry<DEG>(P, p, r);
    float gd[11], wq[4], gbase[kGrtNhtMaxIpd];
    if (!nht_hit_bwd<DEG>(P, r, p, g, features, id, tet, st, gd, wq, gbase)) return;
#pragma unroll
    for (int k = 0; k < 11; ++k) atomicAdd(g_density12 + 12 * (size_t)id + k, gd[k]);
    const int points = P.nht_support == 1 ? 4 : 1;
    for (int k = 0; k < points; ++k)
#pragma unroll
        for (int m = 0; m < kGrtNhtMaxIpd; ++m)
            if (m < P.nht_ipd) atomicAdd(g_features + (size_t)id * P.nht_k + k * P.nht_ipd + m, wq[k] * gbase[m]);
}

// The reference's backward program, round by round.  Without a hit log it serves every ray (render.backward_hit_replay = false, or
// a backward that is not the last logged forward's); with one it serves the rays the forward flagged (kGrtRederiveRay: a chunk could
// not hold all the ghosts of one of their rounds) — or every ray if the log overflowed.  UNI: the frame's packet lists are at hand, a
// round is a window scan of the packet's list (list_round: same candidate sets and order as the tree walk) — with one or two live
// lanes per wave the window is those rays' own, a few dozen entries.
template <int DEG, bool UNI, bool NHT = false>
__global__ __launch_bounds__(64) void grt_trace_bwd_kernel(GrtTraceParams P, GrtBvh bvh, const float4* __restrict__ density12,
                                                           const float* __restrict__ sph, const float* __restrict__ ray_o,
                                                           const float* __restrict__ ray_d, const float* __restrict__ in_rad,
                                                           const float* __restrict__ in_dns, const float* __restrict__ in_hit2,
                                                           const float* __restrict__ g_rad, const float* __restrict__ g_dns,
                                                           const float* __restrict__ g_hit, float* __restrict__ g_density12,
                                                           float* __restrict__ g_sph, const uint32_t* __restrict__ log_state,
                                                           const uint32_t* __restrict__ log_flags, GrtLists lists) {
    __shared__ uint32_t s_stack[UNI ? 1 : kGrtStackDepth];
    __shared__ float s_hit_t[kGrtMaxHits * 64];      // (UNI: a round's staged list entries live here while it is scanned)
    __shared__ uint32_t s_hit_id[kGrtMaxHits * 64];
    static_assert(kGrtMaxHits * 64 * 4 >= 64 * 3 * 16, "the staged list entries must fit the parked hit distances");
    const int lane = threadIdx.x;
    const PixelBlock pb = pixel_block(P.W, P.H);
    if (!pb.inside) return;
    const int px = pb.bx * 8 + (lane & 7), py = pb.by * 8 + (lane >> 3);
    const bool in_image = (px < P.W) && (py < P.H);
    const size_t pix = in_image ? (size_t)py * P.W + px : 0;  // out-of-image lanes shadow pixel 0 and never write
    bool running = in_image;
    if (log_state && log_state[1] == 0u) running = running && (log_flags[pix] & kGrtRederiveRay) != 0u;   // the replay serves the rest
    if (!__any(running)) return;
    const bool handled = running;
    unsigned long long dbg_sig = 0ull;
    uint32_t dbg_n = 0u;
    const RayW r = make_ray(P, ray_o, ray_d, pix);
    float basis[16];
    sh_basis16(P.sph_degree, r.d, basis);
    const int nact = min((P.sph_degree + 1) * (P.sph_degree + 1), P.ncoef);

    BwdRay ray_state;
    NhtBwdRay nht_state;   // (NHT: in_rad / g_rad are the [H,W,ray_dim] feature image and its gradient, sph / g_sph the feature buffer and its gradient)
    NhtTetra tet;
    const float max_hit = in_hit2[2 * pix + 1];
    if constexpr (NHT) {
        nht_state = nht_bwd_ray_init(P, in_image, pix, in_rad, in_dns, in_hit2, g_rad, g_dns, g_hit);
        tet = nht_tetra();
    } else {
        ray_state.rad_fin = load_radiance(P, in_rad, pix);
        ray_state.T_fin = 1.f - in_dns[pix];
        ray_state.depth_fin = in_hit2[2 * pix];
        ray_state.rad_grad = mk3(g_rad[3 * pix], g_rad[3 * pix + 1], g_rad[3 * pix + 2]);
        ray_state.T_grad = -g_dns[pix];
        ray_state.depth_grad = g_hit ? g_hit[pix] : 0.f;
        ray_state.rad = mk3(0.f, 0.f, 0.f);
        ray_state.T = 1.f;
        ray_state.depth = 0.f;
    }
    float tEnter, tExit;
    scene_interval(bvh.scene, r, tEnter, tExit);
    constexpr float eps = 1e-9f;
    float startT = fmaxf(0.f, tEnter - eps);
    const float endT = fminf(max_hit, tExit) + eps;
    uint32_t list_end = 0u, list_start = 0u;
    GrtCone cone = {0.f, 0.f, 1.f, -1.f, 0.f, 1.f, 0.f, 0.f};
    float dmin = 1.f, dmax = 1.f;
    if (UNI) {
        list_start = lists.ranges[2 * (size_t)pb.index];
        list_end = lists.ranges[2 * (size_t)pb.index + 1];
        cone = lists.block_cones[pb.index];
        dmin = __uint_as_float(lists.dir_len_enc[0]); dmax = __uint_as_float(lists.dir_len_enc[1]);
    }
    while (true) {
        running = running && (startT < endT);
        if (!__any(running)) break;
        TraceCounters tc;
        {
            HitBuffer buf;
            if (UNI) list_round<false, kGrtMaxHits>(lists, cone, dmin, dmax, list_end, list_start, r, startT + eps, endT, running, lane,
                                                    reinterpret_cast<float4*>(s_hit_t), buf, tc);
            else trace_round<false>(bvh, r, startT + eps, endT, running, lane, s_stack, buf, tc);
            if (buf.first_id() == 0xFFFFFFFFu) running = false;
            buf.store(s_hit_t, s_hit_id, lane);
        }
#pragma unroll 1
        for (int i = 0; i < kGrtMaxHits; ++i) {
            const uint32_t id = s_hit_id[i * 64 + lane];
            const bool process = running && (id != 0xFFFFFFFFu);
            if (!__any(process)) break;  // ascending list: nothing further for any lane
            if (process) {
                if constexpr (NHT) process_hit_bwd_nht<DEG>(P, r, id, density12, sph, tet, nht_state, g_density12, g_sph);
                else process_hit_bwd<DEG>(P, r, basis, nact, particle_of(P, id), density12, sph, ray_state, g_density12, g_sph);
                startT = fmaxf(startT, s_hit_t[i * 64 + lane]);
                dbg_n++; dbg_sig += (unsigned long long)id * 0x9E3779B97F4A7C15ull + 1ull;
            }
        }
    }
    if (P.bwd_sig && handled) { P.bwd_sig[pix] = dbg_sig; P.bwd_cnt[pix] = dbg_n; }
}

// backward from the forward's hit log — no traversal.
// Every lane walks ITS chunk sequence (processed hits and ghosts in hit-distance order, GrtHitLog) with the state machine of the
// reference's backward program (referenceBwdOptix.cu:123-166): a trace from startT + eps to endT returns the 16 nearest candidates
// whose hit distance lies in that interval and whose box interval touches it, every returned hit is differentiated, startT moves to
// the largest of their distances.  Candidates arrive in ascending order, so "the 16 nearest of a trace" is a running count.
// Gradient traffic.  The reference issues 11 + 48 atomicAdds per hit from the hit's own lane (gaussianParticles.cuh:468-731): 59
// instructions per slot whose 64 lanes address 64 different particles — 64 cache lines each (79 ms at 1 M particles, round 2).
// Round 2 went particle-major: per distinct particle of a slot group the whole wave re-derived the gradient terms and reduce-
// scattered them with DPP.  That pays when many rays of a packet hit the same particle; on the million-particle frames a particle
// covers 1.9 of the packet's 64 rays, and the wave spent 4.9 G instructions on 23 M groups of two lanes (10.2 ms, 77 % VALU-busy).
// Now hit-major with TRANSPOSED atomics: per slot
//   A. every lane advances its ray state over its hit and computes that hit's own 11 geometry-gradient terms and the three
//      clamp-masked colour weights (dL/d radiance * weight), once, in parallel over the lanes, and leaves them in LDS;
//   B. hit by hit (ballot order) the wave turns ONE lane's hit into its 11 + 3 * nact gradient words, lane j holding word j
//      (SH words = the hit ray's basis value, kept in LDS per ray, times the colour weight): one atomic instruction whose lanes
//      address consecutive words of one particle — four cache lines per hit instead of 59 x 1.
constexpr int kTermStride = 15;    // per lane: 11 geometry terms + 3 colour weights, padded to an odd stride (LDS banks)
constexpr int kBasisStride = 17;

// Per-wave tail of the replay kernels: GrtHitLog::state[3] += rays of this wave whose rounds are re-derived, [4] += atomic instructions issued,
// [5] += float words they carried, in units of 16 (what bench.py prices against scripts/atomic_calib.hip; GrtStats reads them back).
__device__ __forceinline__ void replay_bookkeeping(const GrtHitLog& log, bool rederived, uint32_t at_instr, uint32_t at_words, int lane) {
    const unsigned long long red = __ballot(rederived);
    if (lane == 0) {
        if (red) atomicAdd(&log.state[3], (uint32_t)__popcll(red));
        if (at_instr) { atomicAdd(&log.state[4], at_instr); atomicAdd(&log.state[5], (at_words + 8u) >> 4); }
    }
}

template <int DEG, bool GEN>
#ifndef GRT_REPLAY_WAVES
#define GRT_REPLAY_WAVES 3   // round 4: 4 -> 3 waves per SIMD (168 registers, no scratch; at 4 the kernel kept 84 B per lane in scratch): backward 4.2-4.4 -> 3.76 ms; 2: 3.85, 5: 6.3
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GRT_REPLAY_WAVES, GRT_REPLAY_WAVES))) void grt_replay_bwd_kernel(GrtTraceParams P, const float4* __restrict__ density12, const float* __restrict__ sph,
                                                            const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                            const float* __restrict__ in_rad, const float* __restrict__ in_dns,
                                                            const float* __restrict__ in_hit2, const float* __restrict__ g_rad,
                                                            const float* __restrict__ g_dns, const float* __restrict__ g_hit,
                                                            float* __restrict__ g_density12, float* __restrict__ g_sph, GrtHitLog log,
                                                            const float* __restrict__ inst, const float* __restrict__ scene) {
    __shared__ float s_terms[64 * kTermStride], s_basis[64 * kBasisStride];
    if (log.state[1] != 0u) return;  // the log overflowed: the traversal kernel handles this frame
    if (!GEN) P.prim = GRUT_PRIM_INSTANCES;   // (the instantiation for instances: see grt_trace_fwd_kernel)
    const int lane = threadIdx.x;
    const PixelBlock pb = pixel_block(P.W, P.H);
    if (!pb.inside) return;
    const int px = pb.bx * 8 + (lane & 7), py = pb.by * 8 + (lane >> 3);
    const bool in_image = (px < P.W) && (py < P.H);
    const size_t pix = in_image ? (size_t)py * P.W + px : 0;
    const RayW r = make_ray(P, ray_o, ray_d, pix);
    float basis[16];
    sh_basis16(P.sph_degree, r.d, basis);
#pragma unroll
    for (int k = 0; k < 16; ++k) s_basis[lane * kBasisStride + k] = basis[k];
    const int nact = min((P.sph_degree + 1) * (P.sph_degree + 1), P.ncoef);
    const f3 rad_fin = load_radiance(P, in_rad, pix);
    const float T_fin = 1.f - in_dns[pix], depth_fin = in_hit2[2 * pix];
    const f3 rad_grad = mk3(g_rad[3 * pix], g_rad[3 * pix + 1], g_rad[3 * pix + 2]);
    const float T_grad = -g_dns[pix], depth_grad = g_hit ? g_hit[pix] : 0.f;
    f3 rad = mk3(0.f, 0.f, 0.f);
    float T = 1.f, depth = 0.f;
    const bool replayed = in_image && !(log.ray_flags[pix] & kGrtRederiveRay);   // (else: the exact rounds of grt_trace_bwd_kernel serve this ray)
    uint32_t at_instr = 0u, at_words = 0u;
    // the scatter's lane roles: word j of a hit's gradient — SH words first (their row is 3 * ncoef floats), then the 11 packed terms
    const bool sh_lane = lane < 48;
    const int coef = lane / 3, chn = lane - 3 * coef;
    const bool word_used = sh_lane ? (lane < 3 * nact) : (lane < 59);
    const int term_word = sh_lane ? 11 + chn : (lane < 59 ? lane - 48 : 0);
    const int basis_word = sh_lane ? coef : 0;
    float* const row_base = sh_lane ? g_sph + lane : g_density12 + (lane < 59 ? lane - 48 : 0);
    const uint32_t row_stride = sh_lane ? (uint32_t)(3 * P.ncoef) : 12u;
    // the backward program's trace state
    constexpr float eps = 1e-9f;
    float tEnter, tExit;
    scene_interval(scene, r, tEnter, tExit);
    const float endT = fminf(in_hit2[2 * pix + 1], tExit) + eps;
    float bw_start = fmaxf(0.f, tEnter - eps), bw_max = bw_start;   // startT of the running trace, largest hit distance it returned so far
    uint32_t bw_cnt = 0u;                                           // hits the running trace returned so far
    // the forward kept the ghosts of its round q whose box exit reaches back to the tmin of round q - 2 (GhostLog): premise = the running
    // trace never started before that
    float fw_start0 = bw_start, fw_start1 = bw_start, fw_start2 = bw_start, fw_last = bw_start;   // startT of forward rounds q, q - 1, q - 2; last hit distance met
    bool premise_broken = false;
    unsigned long long dbg_sig = 0ull;
    uint32_t dbg_n = 0u;
    const uint32_t block = pb.index;
    __syncthreads();
    for (uint32_t round = 0; round < log.max_rounds; ++round) {
        const uint32_t c = log.table[(size_t)block * log.max_rounds + round];
        if (c == 0xFFFFFFFFu) break;
        const uint32_t* chunk = log.pool + (size_t)c * (kGrtLogSlots * 64) + lane;
        fw_start2 = fw_start1; fw_start1 = fw_start0; fw_start0 = fw_last;
        if (replayed && chunk[0] != 0xFFFFFFFFu && bw_start < fw_start2) premise_broken = true;
#pragma unroll 1
        for (int i = 0; i < kGrtLogSlots; ++i) {
            // ---- A: the lane's own hit ----
            uint32_t id = replayed ? chunk[i * 64] : 0xFFFFFFFFu;
            if (!__any(id != 0xFFFFFFFFu)) break;   // (a chunk's slots are filled from the front on every lane)
            bool contributes = false;
            if (id != 0xFFFFFFFFu) {
                const bool ghost = (id & kGrtGhostBit) != 0u;
                id &= ~kGrtGhostBit;
                // is this candidate offered to the backward's running trace?  (same arithmetic as the traces themselves: candidate())
                const Cand cd = candidate(inst + 12 * (size_t)id, r, -3.0e38f, 3.0e38f, id);
                if (!ghost) fw_last = fmaxf(fw_last, cd.t);
                const float tmin = bw_start + eps;
                if (cd.ok && (cd.t > tmin) && (cd.t < endT) && (cd.tfar >= tmin) && (cd.tnear <= endT)) {
                    bw_max = fmaxf(bw_max, cd.t);
                    if (++bw_cnt == (uint32_t)kGrtMaxHits) { bw_start = bw_max; bw_cnt = 0u; }   // the trace is full: the next one starts behind its last hit
                    dbg_n++; dbg_sig += (unsigned long long)id * 0x9E3779B97F4A7C15ull + 1ull;
                    id = particle_of(P, id);   // from here on: the particle (its rows are the atomics' targets, lanes merge by particle)
                    const Particle p = load_particle(density12, id);
                    const HitGeom g = hit_geometry<DEG>(P, p, r);
                    if (g.accept) {
                        const bool surfel = P.prim == GRUT_PRIM_TRISURFEL;
                        const float pdot = surfel ? -g.gro.z / g.grd.z : -dot(g.grd, g.gro);
                        const f3 grdd = g.grd * pdot;
                        const f3 grds = p.scl * grdd;
                        const float gsq = dot(grds, grds);
                        const float gdist = sqrtf(gsq);
                        const float weight = g.galpha * T;
                        const float nextT = (1.f - g.galpha) * T;
                        depth = fmaf(weight, gdist, depth);
                        const float resHitT = fmaxf(nextT <= P.min_transmittance ? 0.f : (depth_fin - depth) / nextT, 0.f);
                        const float galphaRayHitGrd = (gdist - resHitT) * T * depth_grad;
                        const float resTrm = g.galpha < 0.999999f ? T_fin / (1.f - g.galpha) : T;
                        const float galphaRayDnsGrd = resTrm * -T_grad;
                        const f3 gradu = sh_radiance(P, sph, id, basis);
                        const f3 grad = mk3(fmaxf(gradu.x, 0.f), fmaxf(gradu.y, 0.f), fmaxf(gradu.z, 0.f));
                        rad = rad + grad * weight;
                        f3 resRad = mk3(0.f, 0.f, 0.f);
                        if (!(nextT <= P.min_transmittance)) {
                            const float inT = 1.f / nextT;
                            resRad = mk3(fmaxf((rad_fin.x - rad.x) * inT, 0.f), fmaxf((rad_fin.y - rad.y) * inT, 0.f), fmaxf((rad_fin.z - rad.z) * inT, 0.f));
                        }
                        const float common = galphaRayHitGrd + galphaRayDnsGrd + T * (grad.x - resRad.x) * rad_grad.x + T * (grad.y - resRad.y) * rad_grad.y +
                                             T * (grad.z - resRad.z) * rad_grad.z;
                        T = nextT;
                        contributes = true;
                        // the hit's gradient words (gaussianParticles.cuh:468-731), left in LDS for the scatter
                        float* const tw = s_terms + lane * kTermStride;
                        tw[11] = gradu.x > 0.f ? rad_grad.x * weight : 0.f;
                        tw[12] = gradu.y > 0.f ? rad_grad.y * weight : 0.f;
                        tw[13] = gradu.z > 0.f ? rad_grad.z * weight : 0.f;
                        const float wd = weight * depth_grad;
                        const f3 gscl = p.scl;
                        const f3 grdsRayHitGrd = gsq > 0.f ? grds * (wd / gdist) : mk3(0.f, 0.f, 0.f);
                        const f3 gsclRayHitGrd = grdd * grdsRayHitGrd;
                        const float grdScaledDot = dot(grdsRayHitGrd * gscl, g.grd);
                        f3 grdRayHitGrd, groRayHitGrd;
                        surfel_hit_distance_grads(surfel, g, gscl * grdsRayHitGrd * pdot, pdot, grdScaledDot, grdRayHitGrd, groRayHitGrd);
                        const float gresGrd = p.density * common;
                        const float grayGrd = particle_response_grd<DEG>(g.gray, g.gres, gresGrd);
                        f3 grdGrd, groGrd;
                        gray_distance_grads(surfel, g, grayGrd, grdGrd, groGrd);
                        const f3 groTot = groGrd + groRayHitGrd;
                        const f3 is2 = g.giscl * g.giscl;
                        const f3 gsclGrdGro = mk3(-g.gposcr.x * is2.x, -g.gposcr.y * is2.y, -g.gposcr.z * is2.z) * groTot;
                        const f3 gposcrGrd = g.giscl * groTot;
                        const f3 gposcGrd = mul_cols(p.rotT, gposcrGrd);
                        const float4 gq1 = matmul_bw_quat(g.gposc, gposcrGrd, p.quat);
                        const f3 dn = grdGrd + grdRayHitGrd;
                        const float l2 = dot(g.grdu, g.grdu);
                        f3 grduGrd = mk3(0.f, 0.f, 0.f);
                        if (l2 > 0.f) {
                            const float il = 1.f / sqrtf(l2);
                            grduGrd = dn * il - g.grdu * (il * il * il * dot(dn, g.grdu));
                        }
                        const f3 sclGrd = gsclRayHitGrd + gsclGrdGro + mk3(-g.rdr.x * is2.x, -g.rdr.y * is2.y, -g.rdr.z * is2.z) * grduGrd;
                        const float4 gq2 = matmul_bw_quat(r.d, g.giscl * grduGrd, p.quat);
                        tw[0] = -gposcGrd.x; tw[1] = -gposcGrd.y; tw[2] = -gposcGrd.z; tw[3] = g.gres * common;
                        tw[4] = gq1.x + gq2.x; tw[5] = gq1.y + gq2.y; tw[6] = gq1.z + gq2.z; tw[7] = gq1.w + gq2.w;
                        tw[8] = sclGrd.x; tw[9] = sclGrd.y; tw[10] = sclGrd.z;
                    }
                }
            }
            // ---- B: one atomic instruction per hit, lanes = the words of its gradient ----
            unsigned long long m = __ballot(contributes);
            __syncthreads();
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                const uint32_t pid = (uint32_t)__builtin_amdgcn_readlane((int)id, src);
                // the lanes that hold the same particle in this slot (neighbouring rays meet a particle at the same position of their
                // sequences more often than not) go out together: the kernel is bound by the atomic words it sends
                unsigned long long same = __ballot(contributes && id == pid);
                m &= ~same;
                float v = 0.f;
                while (same) {
                    const int s2 = __ffsll((long long)same) - 1;
                    same &= same - 1;
                    v = fmaf(s_terms[s2 * kTermStride + term_word], sh_lane ? s_basis[s2 * kBasisStride + basis_word] : 1.f, v);
                }
                const bool send = word_used && v != 0.f;
                if (send) atomicAdd(row_base + (size_t)pid * row_stride, v);
                const unsigned long long sent = __ballot(send);   // (scalar bookkeeping: the atomic instructions and words this wave issues)
                at_instr += sent != 0ull;
                at_words += (uint32_t)__popcll(sent);
            }
            __syncthreads();
        }
    }
    if (P.bwd_sig && replayed) { P.bwd_sig[pix] = dbg_sig; P.bwd_cnt[pix] = dbg_n; }
    // rays whose ghost premise failed (see GhostLog): their gradient may miss a hit the reference's backward would have been offered
    const unsigned long long broken = __ballot(premise_broken);
    if (broken && lane == 0) atomicAdd(&log.state[2], (uint32_t)__popcll(broken));
    replay_bookkeeping(log, in_image && !replayed, at_instr, at_words, lane);
}

// The replay backward for neural harmonic features (referenceSlangBwdOptix.cu:70-185): the log walk and the backward program's trace
// state machine of grt_replay_bwd_kernel, per hit nht_hit_bwd instead of the reference pipeline's processHitBwd, and the same hit-major
// transposed atomics: lane j < K carries word j of the particle's feature-row gradient (= barycentric weight x d L / d base feature,
// both left in LDS by the hit's lane), lanes K .. K + 10 its 11 geometric terms.  (K + 11 <= 64.)
constexpr int kNhtTermStride = 33;   // per lane: 11 geometric terms + 4 barycentric weights + 16 base-feature gradients, odd stride
template <int DEG, bool FAST = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void grt_replay_nht_bwd_kernel(
    GrtTraceParams P, const float4* __restrict__ density12, const float* __restrict__ features, const float* __restrict__ ray_o,
    const float* __restrict__ ray_d, const float* __restrict__ in_feat, const float* __restrict__ in_dns, const float* __restrict__ in_hit2,
    const float* __restrict__ g_feat, const float* __restrict__ g_dns, const float* __restrict__ g_hit, float* __restrict__ g_density12,
    float* __restrict__ g_features, GrtHitLog log, const float* __restrict__ inst, const float* __restrict__ scene) {
    __shared__ float s_terms[64 * kNhtTermStride];
    if (log.state[1] != 0u) return;  // the log overflowed: the traversal kernel handles this frame
    const int lane = threadIdx.x;
    const PixelBlock pb = pixel_block(P.W, P.H);
    if (!pb.inside) return;
    const int px = pb.bx * 8 + (lane & 7), py = pb.by * 8 + (lane >> 3);
    const bool in_image = (px < P.W) && (py < P.H);
    const size_t pix = in_image ? (size_t)py * P.W + px : 0;
    const RayW r = make_ray(P, ray_o, ray_d, pix);
    const NhtTetra tet = nht_tetra();
    NhtBwdRay st = nht_bwd_ray_init(P, in_image, pix, in_feat, in_dns, in_hit2, g_feat, g_dns, g_hit);
    const bool replayed = in_image && !(log.ray_flags[pix] & kGrtRederiveRay);
    uint32_t at_instr = 0u, at_words = 0u;
    // the scatter's lane roles
    const int K = P.nht_k, ipd = P.nht_ipd;
    const bool row_lane = lane < K;
    const int kq = row_lane ? lane / ipd : 0, nq = row_lane ? lane - kq * ipd : 0;
    const bool word_used = lane < K + 11;
    float* const row_base = row_lane ? g_features + lane : g_density12 + (word_used ? lane - K : 0);
    const uint32_t row_stride = row_lane ? (uint32_t)K : 12u;
    constexpr float eps = 1e-9f;
    float tEnter, tExit;
    scene_interval(scene, r, tEnter, tExit);
    const float endT = fminf(in_hit2[2 * pix + 1], tExit) + eps;
    float bw_start = fmaxf(0.f, tEnter - eps), bw_max = bw_start;
    uint32_t bw_cnt = 0u;
    float fw_start0 = bw_start, fw_start1 = bw_start, fw_start2 = bw_start, fw_last = bw_start;
    bool premise_broken = false;
    unsigned long long dbg_sig = 0ull;
    uint32_t dbg_n = 0u;
    for (uint32_t round = 0; round < log.max_rounds; ++round) {
        const uint32_t c = log.table[(size_t)pb.index * log.max_rounds + round];
        if (c == 0xFFFFFFFFu) break;
        const uint32_t* chunk = log.pool + (size_t)c * (kGrtLogSlots * 64) + lane;
        fw_start2 = fw_start1; fw_start1 = fw_start0; fw_start0 = fw_last;
        if (replayed && chunk[0] != 0xFFFFFFFFu && bw_start < fw_start2) premise_broken = true;
#pragma unroll 1
        for (int i = 0; i < kGrtLogSlots; ++i) {
            uint32_t id = replayed ? chunk[i * 64] : 0xFFFFFFFFu;
            if (!__any(id != 0xFFFFFFFFu)) break;
            bool contributes = false;
            if (id != 0xFFFFFFFFu) {
                const bool ghost = (id & kGrtGhostBit) != 0u;
                id &= ~kGrtGhostBit;
                const Cand cd = candidate(inst + 12 * (size_t)id, r, -3.0e38f, 3.0e38f, id);
                if (!ghost) fw_last = fmaxf(fw_last, cd.t);
                const float tmin = bw_start + eps;
                if (cd.ok && (cd.t > tmin) && (cd.t < endT) && (cd.tfar >= tmin) && (cd.tnear <= endT)) {
                    bw_max = fmaxf(bw_max, cd.t);
                    if (++bw_cnt == (uint32_t)kGrtMaxHits) { bw_start = bw_max; bw_cnt = 0u; }
                    dbg_n++; dbg_sig += (unsigned long long)id * 0x9E3779B97F4A7C15ull + 1ull;
                    id = particle_of(P, id);   // from here on: the particle (its rows are the atomics' targets, lanes merge by particle)
                    const Particle p = load_particle(density12, id);
                    const HitGeom g = hit_geometry<DEG>(P, p, r);
                    float gd[11], wq[4], gbase[kGrtNhtMaxIpd];
                    contributes = FAST ? nht_hit_bwd_fast<DEG>(P, r, p, g, features, id, tet, st, gd, wq, gbase)
                                       : nht_hit_bwd<DEG>(P, r, p, g, features, id, tet, st, gd, wq, gbase);
                    if (contributes) {
                        float* const tw = s_terms + lane * kNhtTermStride;
#pragma unroll
                        for (int k = 0; k < 11; ++k) tw[k] = gd[k];
#pragma unroll
                        for (int k = 0; k < 4; ++k) tw[11 + k] = wq[k];
#pragma unroll
                        for (int m = 0; m < kGrtNhtMaxIpd; ++m) tw[15 + m] = gbase[m];
                    }
                }
            }
            unsigned long long m = __ballot(contributes);
            __syncthreads();
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                const uint32_t pid = (uint32_t)__builtin_amdgcn_readlane((int)id, src);
                unsigned long long same = __ballot(contributes && id == pid);   // lanes that hold the same particle in this slot go out together
                m &= ~same;
                float v = 0.f;
                while (same) {
                    const int s2 = __ffsll((long long)same) - 1;
                    same &= same - 1;
                    const float* tw = s_terms + s2 * kNhtTermStride;
                    v += row_lane ? tw[11 + kq] * tw[15 + nq] : tw[word_used ? lane - K : 0];
                }
                const bool send = word_used && v != 0.f;
                if (send) atomicAdd(row_base + (size_t)pid * row_stride, v);
                const unsigned long long sent = __ballot(send);
                at_instr += sent != 0ull;
                at_words += (uint32_t)__popcll(sent);
            }
            __syncthreads();
        }
    }
    if (P.bwd_sig && replayed) { P.bwd_sig[pix] = dbg_sig; P.bwd_cnt[pix] = dbg_n; }
    const unsigned long long broken = __ballot(premise_broken);
    if (broken && lane == 0) atomicAdd(&log.state[2], (uint32_t)__popcll(broken));
    replay_bookkeeping(log, in_image && !replayed, at_instr, at_words, lane);
}

// ---------------------------------------------------------------------------------------------
// Hybrid mesh + Gaussian path tracing (BASELINE config 5): threedgrut_playground/src/kernels/cuda/
// playgroundKernel.cu:39-157 (path loop), :159-352 (materials, closest hit), include/playground/kernels/cuda/trace.cuh:175-231,
// 3dgrtTracer.cuh:137-204 (traceVolumetricGS: the forward's k = 16 rounds on a sub-interval, continuing the transmittance).
// Primitive types none / mirror / glass / diffuse; PBR primitives are refused by the host API.  The triangle mesh gets its own
// LBVH (the Gaussian builder's Morton / hierarchy / refit stages over triangle boxes); OptiX's closest-hit is a packet walk
// with a per-lane nearest distance.  Bounce rays of an 8x8 block diverge, the packet walk then visits the union of their
// paths — correct for any ray set, tuned for none (first version: forward only, like the reference).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grt_mesh_aabb_kernel(uint32_t F, const float* __restrict__ vertices, const int32_t* __restrict__ triangles,
                                                            float* __restrict__ aabb, float* __restrict__ slack, uint32_t* __restrict__ scene_enc) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    if (i < F) {
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const float* p = vertices + 3 * (size_t)triangles[3 * (size_t)i + v];
#pragma unroll
            for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], p[k]); hi[k] = fmaxf(hi[k], p[k]); }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {   // a hair of padding: culling must stay conservative, hits are decided on the triangle itself
            const float pad = 1e-5f * (hi[k] - lo[k]) + 1e-6f * (fabsf(lo[k]) + fabsf(hi[k]) + 1.f);
            lo[k] -= pad; hi[k] += pad;
        }
        float* b = aabb + 6 * (size_t)i;
        b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = hi[0]; b[4] = hi[1]; b[5] = hi[2];
        slack[i] = 0.f;
    }
    uint32_t* rep = scene_enc + (blockIdx.x % kSceneReplicas) * kSceneReplicaStride;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float mn = wave_min(lo[k]), mx = wave_max(hi[k]);
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&rep[k], enc_ordered(mn));
            atomicMax(&rep[3 + k], enc_ordered(mx));
        }
    }
}

struct MeshHit {
    float t, u, v;
    uint32_t tri;   // 0xFFFFFFFF: miss
};
// Moeller-Trumbore, written with explicit operation order and no contraction: the CPU checker evaluates bit-identical distances
__device__ __forceinline__ bool tri_intersect(const GrtMeshView& m, uint32_t f, const RayW& r, float tmin, float tmax, float& t, float& u, float& v) {
#pragma clang fp contract(off)
    const int32_t i0 = m.triangles[3 * (size_t)f], i1 = m.triangles[3 * (size_t)f + 1], i2 = m.triangles[3 * (size_t)f + 2];
    const float* p0 = m.vertices + 3 * (size_t)i0; const float* p1 = m.vertices + 3 * (size_t)i1; const float* p2 = m.vertices + 3 * (size_t)i2;
    const float v0x = p0[0], v0y = p0[1], v0z = p0[2];
    const float e1x = p1[0] - v0x, e1y = p1[1] - v0y, e1z = p1[2] - v0z, e2x = p2[0] - v0x, e2y = p2[1] - v0y, e2z = p2[2] - v0z;
    const float pvx = r.d.y * e2z - r.d.z * e2y, pvy = r.d.z * e2x - r.d.x * e2z, pvz = r.d.x * e2y - r.d.y * e2x;
    const float det = e1x * pvx + e1y * pvy + e1z * pvz;
    if (!(fabsf(det) > 1e-20f)) return false;
    const float inv = 1.f / det;
    const float tvx = r.o.x - v0x, tvy = r.o.y - v0y, tvz = r.o.z - v0z;
    u = (tvx * pvx + tvy * pvy + tvz * pvz) * inv;
    if (u < 0.f || u > 1.f) return false;
    const float qx = tvy * e1z - tvz * e1y, qy = tvz * e1x - tvx * e1z, qz = tvx * e1y - tvy * e1x;
    v = (r.d.x * qx + r.d.y * qy + r.d.z * qz) * inv;
    if (v < 0.f || u + v > 1.f) return false;
    t = (e2x * qx + e2y * qy + e2z * qz) * inv;
    return (t > tmin) && (t < tmax);
}
__device__ __forceinline__ MeshHit mesh_closest(const GrtMeshView& m, const RayW& r, float tmin, float tmax, bool active, int lane,
                                                uint32_t* __restrict__ stack) {
    MeshHit best;
    best.t = 3.0e38f; best.u = 0.f; best.v = 0.f; best.tri = 0xFFFFFFFFu;
    if (!__any(active) || m.F == 0) return best;
    auto leaf = [&](uint32_t f, bool lane_on) {
        float t, u, v;
        if (lane_on && tri_intersect(m, f, r, tmin, tmax, t, u, v) && ((t < best.t) || (t == best.t && f < best.tri))) { best.t = t; best.u = u; best.v = v; best.tri = f; }
    };
    if (m.F == 1) { leaf(0u, active); return best; }
    int sp = 0;
    uint32_t cur = 0;
    bool have = true;
    while (true) {
        if (!have) {
            if (sp == 0) break;
            cur = stack[--sp];
        }
        have = false;
        cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
        const cfloat4* nq = reinterpret_cast<const cfloat4*>(reinterpret_cast<uintptr_t>(&m.nodes[cur]));
        const float4 q0 = ld4(nq, 0), q1 = ld4(nq, 1), q2 = ld4(nq, 2), q3 = ld4(nq, 3);
        const uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
        float tn0, tf0, tn1, tf1;
        bool ok0, ok1;
        boxes_hit(q0, q1, q2, r, tn0, tf0, tn1, tf1, ok0, ok1);
        const float bound = fminf(tmax, best.t);
        const bool h0 = active && (c0 != kGrtNoChild) && ok0 && (tf0 >= tmin) && (tn0 <= bound);
        const bool h1 = active && (c1 != kGrtNoChild) && ok1 && (tf1 >= tmin) && (tn1 <= bound);
        bool a0 = __any(h0), a1 = __any(h1);
        if (a0 && (c0 & kGrtLeafBit)) { leaf(c0 & ~kGrtLeafBit, h0); a0 = false; }
        if (a1 && (c1 & kGrtLeafBit)) { leaf(c1 & ~kGrtLeafBit, h1); a1 = false; }
        if (a0 && a1) {
            const int v0 = __popcll(__ballot(h0 && (!h1 || tn0 <= tn1))), v1 = __popcll(__ballot(h1 && (!h0 || tn1 < tn0)));
            const bool first0 = v0 >= v1;
            if (lane == 0) stack[sp] = first0 ? c1 : c0;
            sp++;
            cur = first0 ? c0 : c1;
            have = true;
        } else if (a0 || a1) {
            cur = a0 ? c0 : c1;
            have = true;
        }
    }
    return best;
}

// traceVolumetricGS (3dgrtTracer.cuh:137-204) on [tmin, tmax] of ray r for the lanes with `active`: continues T and rad
// REFINE (LISTS only): this call is the LAST scan of the packet's list by a set of rays that contains every ray of any later scan —
// list_round may then keep exact per-entry intervals (see there).  The hybrid tracer scans the list twice in its first iteration
// (the diffuse rays, then all rays): only the second scan refines.
#ifndef GRT_LANE_WALK
#define GRT_LANE_WALK 0   // 1: bounced segments walk the tree per lane (trace_round_lanes).  Built and measured in round 6: bit-identical, SLOWER - config 5
                          // 98.7 vs 81 ms: the sparse scene's rays never fill their buffers, so a single ray's path through 2 M particles is thousands of
                          // nodes long, and a per-lane step is four divergent 16-byte gathers at two waves per SIMD instead of one scalar fetch
#endif
template <int DEG, bool LISTS = false, bool REFINE = false>
__device__ __forceinline__ void trace_segment(const GrtTraceParams& P, const GrtBvh& bvh, const float4* __restrict__ density12,
                                              const float* __restrict__ sph, const RayW& r, float tmin, float tmax, bool active, int lane,
                                              uint32_t* __restrict__ s_stack, float* __restrict__ s_hit_t, uint32_t* __restrict__ s_hit_id, float& T,
                                              f3& rad, const GrtLists* lists = nullptr, const GrtCone* cone = nullptr, float dmin = 1.f,
                                              float dmax = 1.f, uint32_t list_begin = 0u, uint32_t list_end = 0u) {
    constexpr float eps = 1e-9f;
    float t0, t1;
    scene_interval(bvh.scene, r, t0, t1);
    t0 = fmaxf(t0, tmin);
    t1 = fminf(t1, tmax);
    float tLast = fmaxf(0.f, t0 - eps);
    float basis[16];
    sh_basis16(P.sph_degree, r.d, basis);
    TraceCounters tc;
    bool running = active;
    uint32_t list_start = list_begin;   // LISTS: a segment scans the packet's list from its beginning (see list_round)
    float list_span = 3.0e38f;
    while (true) {
        running = running && (tLast <= t1) && (T > P.min_transmittance);
        if (!__any(running)) break;
        {
            HitBufferT<kGrtMaxHits> buf;
            if (LISTS) list_round<false, kGrtMaxHits, false, REFINE>(*lists, *cone, dmin, dmax, list_end, list_start, r, tLast + eps, t1 + eps, running, lane,
                                                                     reinterpret_cast<float4*>(s_hit_t), buf, tc, nullptr, s_hit_id + kGrtMaxGhosts * 64, &list_span,
                                                                     P.list_mark);
            else {
                // rays with their own origins (bounced segments): one walk per lane, its stack in the LDS words of the round's hit arrays - which
                // hold nothing while the round is gathered (32 levels per lane; deeper: the round is repeated by the packet walk)
                static_assert(sizeof(float) == 4 && kGrtMaxHits * 64 * 2 >= 32 * 64, "the per-lane stacks live in s_hit_t + s_hit_id");
                bool done = false;
                if (GRT_LANE_WALK)   // (s_hit_id = s_hit_t + 16 x 64 words: the caller's ONE array, grt_hybrid_kernel)
                    done = trace_round_lanes<kGrtMaxHits>(bvh, r, tLast + eps, t1 + eps, running, lane, reinterpret_cast<uint32_t*>(s_hit_t), 32, buf);
                if (!done) trace_round<false, kGrtMaxHits>(bvh, r, tLast + eps, t1 + eps, running, lane, s_stack, buf, tc);
                __syncthreads();   // single-wave workgroup: the stacks' last reads before the hit arrays are written
            }
            buf.store(s_hit_t, s_hit_id, lane);
        }
        if (s_hit_id[lane] == 0xFFFFFFFFu) running = false;
#pragma unroll 1
        for (int i = 0; i < kGrtMaxHits; ++i) {
            const uint32_t id = s_hit_id[i * 64 + lane];
            const bool process = running && (id != 0xFFFFFFFFu) && (T > P.min_transmittance);
            if (!__any(process)) break;   // ascending lists: nothing further for any lane
            if (process) {
                const uint32_t pid = particle_of(P, id);   // (trihexa: three proxies per particle - the tree and the buffers are keyed by proxy)
                const Particle p = load_particle(density12, pid);
                const HitGeom g = hit_geometry<DEG>(P, p, r);
                if (g.accept) {
                    const float weight = g.galpha * T;
                    const f3 u = sh_radiance(P, sph, pid, basis);
                    rad = rad + mk3(fmaxf(u.x, 0.f), fmaxf(u.y, 0.f), fmaxf(u.z, 0.f)) * weight;
                    T *= (1.f - g.galpha);
                }
                tLast = fmaxf(tLast, s_hit_t[i * 64 + lane]);
            }
        }
    }
}

__device__ __forceinline__ f3 safe_normalize3(f3 v) {
    const float l = dot(v, v);
    return l > 0.f ? v * (1.f / sqrtf(l)) : v;
}
__device__ __forceinline__ f3 mirror_dir(f3 d, f3 n) {   // playgroundKernel.cu:190-198
    const f3 nn = dot(d, n) < 0.f ? n : n * -1.f;
    return safe_normalize3(d - nn * (2.f * dot(nn, d)));
}
__device__ __forceinline__ bool refract_dir(f3& out, f3 d, f3 n, float etai_over_etat) {   // :159-188
    float ri;
    if (dot(d, n) < 0.f) ri = 1.f / etai_over_etat;
    else { ri = etai_over_etat; n = n * -1.f; }
    const float cos_theta = fminf(dot(d * -1.f, n), 1.f);
    const float sin_theta = sqrtf(1.f - cos_theta * cos_theta);
    if (!(ri * sin_theta <= 1.f)) return false;
    const f3 perp = (d + n * cos_theta) * ri;
    const f3 par = n * (-sqrtf(fabsf(1.f - dot(perp, perp))));
    out = safe_normalize3(perp + par);
    return true;
}

// ---- materials of the playground (materials.cuh), random streams (rng.cuh), textures, environment ------------------------------------
// tex2D with the modes of playground/cutexture.h:54-60: normalised coordinates, clamp-to-edge, bilinear (float weights), float texels
__device__ __forceinline__ float4 tex_fetch(const GrtTexture& t, float u, float v) {
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (t.data) {
        const float x = u * (float)t.width - 0.5f, y = v * (float)t.height - 0.5f;
        const float fx = floorf(x), fy = floorf(y);
        const float ax = x - fx, ay = y - fy;
        const int x0 = min(max((int)fx, 0), t.width - 1), x1 = min(max((int)fx + 1, 0), t.width - 1);
        const int y0 = min(max((int)fy, 0), t.height - 1), y1 = min(max((int)fy + 1, 0), t.height - 1);
        for (int c = 0; c < t.channels && c < 4; ++c) {
            const float t00 = t.data[((size_t)y0 * t.width + x0) * t.channels + c], t10 = t.data[((size_t)y0 * t.width + x1) * t.channels + c];
            const float t01 = t.data[((size_t)y1 * t.width + x0) * t.channels + c], t11 = t.data[((size_t)y1 * t.width + x1) * t.channels + c];
            o[c] = (1.f - ay) * ((1.f - ax) * t00 + ax * t10) + ay * ((1.f - ax) * t01 + ax * t11);
        }
    }
    return make_float4(o[0], o[1], o[2], o[3]);
}
__device__ __forceinline__ uint32_t pg_tea16(uint32_t val0, uint32_t val1) {   // tea<16>, rng.cuh:21-35
    uint32_t v0 = val0, v1 = val1, s0 = 0u;
#pragma unroll 1
    for (int n = 0; n < 16; ++n) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}
__device__ __forceinline__ float pg_rnd(uint32_t& prev) {   // lcg + rnd, rng.cuh:38-54
    prev = 1664525u * prev + 1013904223u;
    return (float)(prev & 0x00FFFFFFu) / (float)0x01000000;
}
__device__ __forceinline__ f3 pg_rnd_pcg3d(uint32_t x, uint32_t y, uint32_t z) {   // rng.cuh:62-76
    x = x * 1664525u + 1013904223u; y = y * 1664525u + 1013904223u; z = z * 1664525u + 1013904223u;
    x += y * z; y += z * x; z += x * y;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16;
    x += y * z; y += z * x; z += x * y;
    const float k = 2.3283064365386963e-10f;   // 1 / float(0xffffffff) = 2^-32
    return mk3((float)x * k, (float)y * k, (float)z * k);
}
constexpr float kPbrEps = 1e-6f, kPgPi = 3.141592654f;
__device__ __forceinline__ f3 pg_normalize(f3 v) {   // materials.cuh:79-82 (threshold on the squared norm, not safe_normalize)
    const float n = dot(v, v);
    return (n > kPbrEps) ? v * (1.f / sqrtf(n)) : v;
}
__device__ __forceinline__ float pg_pdot(f3 a, f3 b) { return fminf(1.f, fmaxf(0.f, dot(a, b))); }   // positive_dot
struct PgHit {
    uint32_t tri;
    float bu, bv;
};
__device__ __forceinline__ void pg_tex_coords(const GrtMeshView& m, const PgHit& h, float& u, float& v) {   // materials.cuh:41-48
    u = v = 0.f;
    if (!m.mat_uv) return;
    const float* uv = m.mat_uv + 6 * (size_t)h.tri;
    const float w0 = 1.f - h.bu - h.bv;
    u = w0 * uv[0] + h.bu * uv[2] + h.bv * uv[4];
    v = w0 * uv[1] + h.bu * uv[3] + h.bv * uv[5];
}
__device__ __forceinline__ const GrtMaterial& pg_material(const GrtMeshView& m, uint32_t tri) {
    const uint32_t id = m.mat_id ? (uint32_t)m.mat_id[tri] : 0u;
    return m.materials[id < m.num_materials ? id : 0u];
}
__device__ __forceinline__ f3 pg_diffuse_color(const GrtMeshView& m, const PgHit& h, uint32_t opts, f3 ray_d, f3 normal) {   // get_diffuse_color, :34-66
    const GrtMaterial& mat = pg_material(m, h.tri);
    float u, v;
    pg_tex_coords(m, h, u, v);
    f3 diffuse = mk3(mat.diffuse_factor[0], mat.diffuse_factor[1], mat.diffuse_factor[2]);
    if (mat.diffuse.data && !(opts & 4u)) {
        const float4 t = tex_fetch(mat.diffuse, u, v);
        diffuse = mk3(t.x, t.y, t.z) * diffuse;
    }
    return diffuse * fabsf(dot(ray_d, normal));
}
__device__ __forceinline__ f3 pg_normal_space(const GrtMeshView& m, const PgHit& h, f3 normal, f3 dir) {   // compute_normal_space, :84-135
    const int32_t i0 = m.triangles[3 * (size_t)h.tri], i1 = m.triangles[3 * (size_t)h.tri + 1], i2 = m.triangles[3 * (size_t)h.tri + 2];
    f3 tangent;
    if (m.vhas_tangents && m.vhas_tangents[i0] && m.vhas_tangents[i1] && m.vhas_tangents[i2]) {   // get_smooth_tangent
        const float* t0 = m.vtangents + 3 * (size_t)i0; const float* t1 = m.vtangents + 3 * (size_t)i1; const float* t2 = m.vtangents + 3 * (size_t)i2;
        const float w0 = 1.f - h.bu - h.bv;
        tangent = mk3(w0 * t0[0] + h.bu * t1[0] + h.bv * t2[0], w0 * t0[1] + h.bu * t1[1] + h.bv * t2[1], w0 * t0[2] + h.bu * t1[2] + h.bv * t2[2]);
        const float len = sqrtf(dot(tangent, tangent));
        tangent = mk3(tangent.x / len, tangent.y / len, tangent.z / len);
    } else if (fabsf(normal.x) > fabsf(normal.z)) tangent = mk3(-normal.y, normal.x, 0.f);
    else tangent = mk3(0.f, -normal.z, normal.y);
    tangent = pg_normalize(tangent);
    const f3 bitangent = pg_normalize(cross(normal, tangent));
    return pg_normalize(mk3(tangent.x * dir.x + bitangent.x * dir.y + normal.x * dir.z, tangent.y * dir.x + bitangent.y * dir.y + normal.y * dir.z,
                            tangent.z * dir.x + bitangent.z * dir.y + normal.z * dir.z));
}
__device__ __forceinline__ f3 pg_sample_specular(const GrtMeshView& m, const PgHit& h, f3 normal, float theta_seed, float phi_seed, float roughness) {   // :151-164
    const float alpha = roughness * roughness;
    const float theta = acosf(sqrtf((1.f - theta_seed) / (1.f + (alpha * alpha - 1.f) * theta_seed)));
    const float phi = 2.f * kPgPi * phi_seed;
    return pg_normal_space(m, h, normal, mk3(sinf(theta) * cosf(phi), sinf(theta) * sinf(phi), cosf(theta)));
}
__device__ __forceinline__ float pg_g1(float ndv, float roughness) {   // geometry_schlick_ggx, :196-201
    const float alpha = 0.5f * roughness * roughness;
    return ndv / fmaxf(ndv * (1.f - alpha) + alpha, kPbrEps);
}
__device__ __forceinline__ f3 pg_fresnel(float cosine, f3 f0) {   // fresnel_schlick, :212-215
    const float p = powf(1.f - cosine, 5.f);
    return mk3(f0.x + (1.f - f0.x) * p, f0.y + (1.f - f0.y) * p, f0.z + (1.f - f0.z) * p);
}
struct PgPbr {   // the fields of HybridRayPayload the PBR branch writes
    f3 bsdf, emissive;
    uint32_t pbr_bounces, rnd_seed;
};
// sampled_cook_torrance_brdf + sampled_microfacet_brdf (materials.cuh:231-440): material fetch (factors x textures), normal map, alpha
// test, one of the three sampled lobes (transmission / diffuse / specular) chosen by the pixel's random stream
__device__ __noinline__ void pg_cook_torrance(const GrtMeshView& m, const PgHit& h, uint32_t opts, f3 ray_d, f3 normal, uint32_t px, uint32_t py,
                                              uint32_t frame, PgPbr& st, f3& new_dir) {
    const f3 wo = pg_normalize(ray_d * -1.f);
    const GrtMaterial& mat = pg_material(m, h.tri);
    const bool notex = (opts & 4u) != 0u;
    float u, v;
    pg_tex_coords(m, h, u, v);
    f3 base = mk3(mat.diffuse_factor[0], mat.diffuse_factor[1], mat.diffuse_factor[2]);
    float alpha = mat.diffuse_factor[3];
    if (mat.diffuse.data && !notex) {
        const float4 t = tex_fetch(mat.diffuse, u, v);
        base = mk3(t.x, t.y, t.z) * base;
        alpha *= t.w;
    }
    f3 emissive = mk3(mat.emissive_factor[0], mat.emissive_factor[1], mat.emissive_factor[2]);
    if (mat.emissive.data && !notex) {
        const float4 t = tex_fetch(mat.emissive, u, v);
        emissive = mk3(t.x, t.y, t.z) * emissive;
    }
    float metalness = mat.metallic_factor, roughness = mat.roughness_factor;
    if (mat.metallic_roughness.data && !notex) {
        const float4 t = tex_fetch(mat.metallic_roughness, u, v);
        metalness = t.x * mat.metallic_factor;
        roughness = t.y * mat.roughness_factor;
    }
    if (mat.normal.data && !notex) {
        const float4 t = tex_fetch(mat.normal, u, v);
        normal = pg_normal_space(m, h, normal, mk3(t.x, t.y, t.z));
    }
    bool pass = true;   // alpha_test, :166-181
    if (mat.alpha_mode == 1u) pass = alpha > pg_rnd(st.rnd_seed);
    else if (mat.alpha_mode == 2u) pass = alpha > mat.alpha_cutoff;
    if (!pass) { new_dir = ray_d; return; }
    const float transmission = mat.transmission_factor, ior = mat.ior;
    const f3 rnd = pg_rnd_pcg3d(px, py, frame + st.pbr_bounces);
    const float phi_seed = rnd.x, theta_seed = rnd.y, ray_prob = rnd.z;
    const float fr = 0.5f;
    f3 f0 = mk3(0.16f * fr * fr, 0.16f * fr * fr, 0.16f * fr * fr);
    f0 = mk3(f0.x + metalness * (base.x - f0.x), f0.y + metalness * (base.y - f0.y), f0.z + metalness * (base.z - f0.z));
    f3 out, L;
    if ((ray_prob < 0.5f) && ((2.f * ray_prob) < transmission)) {   // transmissive
        const float front = dot(wo, normal);
        const f3 fn = front >= 0.f ? normal : normal * -1.f;
        const float eta = front >= 0.f ? 1.f / ior : ior;
        const f3 H = pg_sample_specular(m, h, fn, theta_seed, phi_seed, roughness);
        const f3 wi = wo * -1.f;   // pbr_refract(-wo, H, eta), :217-222
        const float ndi = dot(H, wi);
        const float kk = 1.f - eta * eta * (1.f - ndi * ndi);
        L = (kk < 0.f) ? mk3(0.f, 0.f, 0.f) : (wi * eta - H * (eta * ndi + sqrtf(kk)));
        const float ndo = pg_pdot(fn, wo), ndl = pg_pdot(fn * -1.f, L), ndh = pg_pdot(fn, H), odh = pg_pdot(wo, H);
        const f3 Fv = pg_fresnel(odh, f0);
        const float G = pg_g1(ndo, roughness) * pg_g1(ndl, roughness);
        const float k = G * odh / fmaxf(ndh * ndo, 0.001f);
        out = mk3(base.x * (1.f - Fv.x) * k, base.y * (1.f - Fv.y) * k, base.z * (1.f - Fv.z) * k);
    } else if ((ray_prob < 0.5f) && ((2.f * ray_prob) >= transmission)) {   // diffuse (importance_sample_diffuse_ggx, :137-149)
        const float theta = asinf(theta_seed), phi = 2.f * kPgPi * phi_seed;
        L = pg_normal_space(m, h, normal, mk3(sinf(theta) * cosf(phi), sinf(theta) * sinf(phi), cosf(theta)));
        const f3 H = pg_normalize(wo + L);
        const f3 Fv = pg_fresnel(pg_pdot(wo, H), f0);
        const float nm = 1.f - metalness;
        out = mk3((1.f - Fv.x) * nm * base.x, (1.f - Fv.y) * nm * base.y, (1.f - Fv.z) * nm * base.z);
    } else {   // specular
        const f3 H = pg_sample_specular(m, h, normal, theta_seed, phi_seed, roughness);
        const f3 mwo = wo * -1.f;
        L = mwo - H * (2.f * dot(H, mwo));
        const float ndo = pg_pdot(normal, wo), ndh = pg_pdot(normal, H), ndl = pg_pdot(normal, L), odh = pg_pdot(wo, H);
        const f3 Fv = pg_fresnel(odh, f0);
        const float G = pg_g1(ndo, roughness) * pg_g1(ndl, roughness);
        out = Fv * (G * odh / fmaxf(ndh * ndo, 0.001f));
    }
    new_dir = L;
    st.pbr_bounces += 1u;
    out = out * 2.f;   // compensates for splitting diffuse and specular
    st.bsdf = mk3(fmaxf(out.x, 0.f), fmaxf(out.y, 0.f), fmaxf(out.z, 0.f));
    st.emissive = emissive;
}
__device__ __forceinline__ f3 pg_background(const GrtMeshView& m, f3 d) {   // getBackgroundColor, trace.cuh:233-257
    const float kPi = 3.14159265358979323846f;
    const float rotY = m.envmap_offset[0] * 2.0f * kPi, rotX = 2.0f * m.envmap_offset[1] * kPi;
    const float cy = cosf(rotY), sy = sinf(rotY), cx = cosf(rotX), sx = sinf(rotX);
    const f3 r1 = mk3(d.x * cy - d.z * sy, d.y, d.x * sy + d.z * cy);
    const f3 r2 = mk3(r1.x, r1.y * cx - r1.z * sx, r1.y * sx + r1.z * cx);
    const float theta = atan2f(r2.x, r2.z);
    const float phi = kPi * 0.5f - acosf(fminf(1.f, fmaxf(-1.f, r2.y)));
    const float u = (theta + kPi) * (0.5f * 0.318309886183790671538f);
    const float v = 0.5f * (1.0f + sinf(phi));
    const float4 t = tex_fetch(m.envmap, u, v);
    return mk3(t.x, t.y, t.z);
}

// GEN = false: render.primitive_type instances with fp32 SH rows (the playground's default) - every `prim` comparison of the candidate test
// and of the per-hit code folds away, as in the training forward's default instantiation
template <int DEG, bool GEN = true>
#ifndef GRT_HYBRID_WAVES
#define GRT_HYBRID_WAVES 0   // 0: the allocator's own choice (219 registers, 2 waves per SIMD)
#endif
__global__ __launch_bounds__(64)
#if GRT_HYBRID_WAVES > 0
__attribute__((amdgpu_waves_per_eu(GRT_HYBRID_WAVES, GRT_HYBRID_WAVES)))
#endif
void grt_hybrid_kernel(GrtTraceParams P, GrtBvh bvh, GrtMeshView mesh, GrtHybridParams hp,
                                                        const float4* __restrict__ density12, const float* __restrict__ sph,
                                                        const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                        const float* __restrict__ ray_max_t, float* __restrict__ out_rgb,
                                                        float* __restrict__ out_alpha, float* __restrict__ out_last_ray,
                                                        uint32_t* __restrict__ out_bounces, GrtLists lists) {
    __shared__ uint32_t s_stack[kGrtStackDepth];
    __shared__ float s_hits[2 * kGrtMaxHits * 64];   // one array: the per-lane walk's stacks span both halves (trace_segment)
    float* s_hit_t = s_hits;
    uint32_t* s_hit_id = reinterpret_cast<uint32_t*>(s_hits + kGrtMaxHits * 64);
    if (!GEN) { P.prim = GRUT_PRIM_INSTANCES; P.sph_half = 0; }
    const int lane = threadIdx.x;
    const PixelBlock pb = pixel_block(P.W, P.H);
    if (!pb.inside) return;
    const int px = pb.bx * 8 + (lane & 7), py = pb.by * 8 + (lane >> 3);
    const bool in_image = (px < P.W) && (py < P.H);
    const size_t pix = in_image ? (size_t)py * P.W + px : 0;
    RayW r = make_ray(P, ray_o, ray_d, pix);
    const float ray_t_max = ray_max_t ? ray_max_t[pix] : 1e30f;
    const bool gaussians = !(hp.opts & 2u);        // PGRNDRenderDisableGaussianTracing
    const bool trace_gs = gaussians && bvh.N > 0;   // (an empty cloud has no tree to walk)
    // payload (playgroundKernel.cu:51-68) and the ray's running volumetric state (RayData: radiance, density = 1 - T)
    f3 accC = mk3(0.f, 0.f, 0.f), direct = accC, thr = mk3(1.f, 1.f, 1.f), rad = accC;
    float accA = 0.f, T = 1.f;
    uint32_t bounces = 0u, timeout = 0u;
    PgPbr pbr;
    pbr.pbr_bounces = 0u;
    pbr.rnd_seed = pg_tea16((uint32_t)P.W * (uint32_t)py + (uint32_t)px, hp.frame_number);
    pbr.bsdf = mk3(1.f, 1.f, 1.f); pbr.emissive = mk3(0.f, 0.f, 0.f);
    bool missed = false, timed_out = false;
    int state = 0;   // PlaygroundTraceState: 0 primitives pass, 1 Gaussians pass, 2 terminate
    f3 lastO = r.o, lastD = r.d;
    // the primary segment of every path (all rays of the frame start at one point) scans the packet's candidate list instead of walking the
    // tree; after the first surface the rays have their own origins
    const bool use_lists = lists.ranges != nullptr;
    GrtCone cone = {0.f, 0.f, 1.f, -1.f, 0.f, 1.f, 0.f, 0.f};
    float dmin = 1.f, dmax = 1.f;
    uint32_t list_begin = 0u, list_end = 0u;
    if (use_lists) {
        list_begin = lists.ranges[2 * (size_t)pb.index]; list_end = lists.ranges[2 * (size_t)pb.index + 1];
        cone = lists.block_cones[pb.index];
        dmin = __uint_as_float(lists.dir_len_enc[0]); dmax = __uint_as_float(lists.dir_len_enc[1]);
    }
    bool primary = true;
    while (true) {
        const bool go = in_image && !missed && !timed_out && (sqrtf(dot(thr, thr)) > 0.0001f) && (accA < 0.995f) && (pbr.pbr_bounces < hp.max_pbr_bounces) &&
                        (bounces < 32u) && (state != 2);
        if (!__any(go)) break;
        if (go) { pbr.bsdf = mk3(1.f, 1.f, 1.f); pbr.emissive = mk3(0.f, 0.f, 0.f); state = 0; }
        // traceMesh + __closesthit__ch / __miss__ms
        const MeshHit mh = mesh_closest(mesh, r, 1e-5f, 1e5f, go, lane, s_stack);
        const bool hit = go && mh.tri != 0xFFFFFFFFu;
        if (go && !hit) missed = true;
        float hit_t = mh.t;
        f3 new_dir = mk3(0.f, 0.f, 0.f), normal = mk3(0.f, 0.f, 1.f);
        int type = 0;
        PgHit ph = {0u, 0.f, 0.f};
        if (hit) {
            ph.tri = mh.tri; ph.bu = mh.u; ph.bv = mh.v;
            const int32_t i0 = mesh.triangles[3 * (size_t)mh.tri], i1 = mesh.triangles[3 * (size_t)mh.tri + 1], i2 = mesh.triangles[3 * (size_t)mh.tri + 2];
            if (hp.opts & 1u) {   // interpolated vertex normals (getSmoothNormal, playgroundKernel.cu:253-274)
                const float w0 = 1.f - mh.u - mh.v;
                const float* n0 = mesh.vnormals + 3 * (size_t)i0; const float* n1 = mesh.vnormals + 3 * (size_t)i1; const float* n2 = mesh.vnormals + 3 * (size_t)i2;
                normal = mk3(w0 * n0[0] + mh.u * n1[0] + mh.v * n2[0], w0 * n0[1] + mh.u * n1[1] + mh.v * n2[1], w0 * n0[2] + mh.u * n1[2] + mh.v * n2[2]);
                const float len = sqrtf(dot(normal, normal));
                normal = mk3(normal.x / len, normal.y / len, normal.z / len);
            } else {              // getHardNormal (:276-286)
                const float* p0 = mesh.vertices + 3 * (size_t)i0; const float* p1 = mesh.vertices + 3 * (size_t)i1; const float* p2 = mesh.vertices + 3 * (size_t)i2;
                normal = safe_normalize3(cross(mk3(p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]), mk3(p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2])));
            }
            type = mesh.prim_type[mh.tri];
            state = (type == 3) ? 2 : 1;
            if (type == 1) { new_dir = mirror_dir(r.d, normal); bounces++; }
            else if (type == 2) {
                const float ior = mesh.refr[mh.tri] / 1.0003f;
                if (refract_dir(new_dir, r.d, normal, ior)) hit_t += 1e-5f;
                else { new_dir = mirror_dir(r.d, normal); bounces++; }
            } else if (type == 4) {   // handlePBR (:226-233)
                pg_cook_torrance(mesh, ph, hp.opts, r.d, normal, (uint32_t)px, (uint32_t)py, hp.frame_number, pbr, new_dir);
                const float len = sqrtf(dot(new_dir, new_dir));
                new_dir = mk3(new_dir.x / len, new_dir.y / len, new_dir.z / len);
            } else if (type != 3) new_dir = r.d;
        }
        // handleDiffuse (:235-251): the Gaussians in front of the surface, then the opaque surface itself
        const bool diffuse = hit && type == 3;
        if (__any(diffuse)) {
            const float T0 = T;
            const f3 rad0 = rad;
            if (trace_gs) {
                if (primary && use_lists) trace_segment<DEG, true>(P, bvh, density12, sph, r, 1e-9f, hit_t, diffuse, lane, s_stack, s_hit_t, s_hit_id, T, rad, &lists, &cone, dmin, dmax, list_begin, list_end);
                else trace_segment<DEG>(P, bvh, density12, sph, r, 1e-9f, hit_t, diffuse, lane, s_stack, s_hit_t, s_hit_id, T, rad);
            }
            if (diffuse) {
                accC = accC + (rad - rad0);
                accA += (1.f - T) - (1.f - T0);
                const f3 dc = pg_diffuse_color(mesh, ph, hp.opts, r.d, normal);
                const float sa = 1.f - accA;
                accC = accC + dc * sa;
                accA += sa;
            }
        }
        // the Gaussians between the ray origin and the surface (or the ray's end); traceGaussians sets the trace state itself (trace.cuh:209)
        {
            const float next_t = missed ? ray_t_max : hit_t;
            const float T0 = T;
            const f3 rad0 = rad;
            if (trace_gs) {
                if (primary && use_lists) trace_segment<DEG, true, true>(P, bvh, density12, sph, r, 1e-9f, next_t, go, lane, s_stack, s_hit_t, s_hit_id, T, rad, &lists, &cone, dmin, dmax, list_begin, list_end);
                else trace_segment<DEG>(P, bvh, density12, sph, r, 1e-9f, next_t, go, lane, s_stack, s_hit_t, s_hit_id, T, rad);
            }
            primary = false;
            if (go) {
                if (gaussians) state = 1;
                const f3 radiance = rad - rad0;
                const float density = (1.f - T) - (1.f - T0);
                accA += density * (1.f - accA);
                accC = accC + thr * radiance;
                direct = direct + radiance;
                thr = thr * (1.f - density);
                accC = accC + thr * (direct + pbr.emissive);
                thr = thr * pbr.bsdf;
                lastO = r.o; lastD = r.d;
                if (++timeout > 1000u) timed_out = true;
                if (hit) {   // next ray of the path
                    r.o = r.o + r.d * hit_t;
                    r.d = new_dir;
                    r.inv = mk3(safe_rcp(r.d.x), safe_rcp(r.d.y), safe_rcp(r.d.z));
                }
            }
        }
    }
    if (!in_image) return;
    direct = direct + pg_background(mesh, lastD);
    thr = thr * (1.f - accA);
    accC = accC + thr * direct;
    accA = fminf(fmaxf(accA, 0.f), 1.f);
    out_rgb[3 * pix] = accC.x; out_rgb[3 * pix + 1] = accC.y; out_rgb[3 * pix + 2] = accC.z;
    out_alpha[pix] = accA;
    if (out_last_ray) {
        float* q = out_last_ray + 6 * pix;
        q[0] = lastO.x; q[1] = lastO.y; q[2] = lastO.z; q[3] = lastD.x; q[4] = lastD.y; q[5] = lastD.z;
    }
    if (out_bounces) out_bounces[pix] = bounces;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
void grt_launch_proxies(hipStream_t s, const GrtBuildParams& P, const float* pos, const float* rot, const float* scl, const float* dns,
                        float* inst, float* aabb, float* slack, uint32_t* scene_enc, float* box8) {
    hipLaunchKernelGGL(grt_scene_init_kernel, dim3(1), dim3(64), 0, s, scene_enc);
    hipLaunchKernelGGL(grt_proxy_kernel, dim3(div_up(P.N, 256)), dim3(256), 0, s, P, pos, rot, scl, dns, inst, aabb, slack, scene_enc, box8);
}
size_t grt_scene_enc_bytes() { return (size_t)kSceneReplicas * kSceneReplicaStride * sizeof(uint32_t); }
void grt_launch_morton(hipStream_t s, uint32_t N, const float* aabb, const uint32_t* scene_enc, float* scene, uint32_t* codes, uint32_t* ids) {
    hipLaunchKernelGGL(grt_morton_kernel, dim3(div_up(N, 256)), dim3(256), 0, s, N, aabb, scene_enc, scene, codes, ids);
}
void grt_launch_hierarchy(hipStream_t s, uint32_t N, const uint32_t* sorted_codes, const uint32_t* sorted_ids, GrtNode* nodes) {
    if (N > 1) hipLaunchKernelGGL(grt_hierarchy_kernel, dim3(div_up(N - 1, 256)), dim3(256), 0, s, N, sorted_codes, sorted_ids, nodes);
}
void grt_launch_refit(hipStream_t s, uint32_t N, const float* aabb, const float* slack, GrtNode* nodes, uint8_t* done, uint32_t* todo) {
    // a radix tree over 30-bit keys + 32 index bits (duplicates) is at most 62 levels deep: kGrtRefitPasses level-synchronous launches over all
    // nodes, then the few nodes above that in one workgroup (`todo`: N + 1 words, [0] zeroed here)
    const uint32_t passes = N <= 2 ? 1u : (todo ? kGrtRefitPasses : (uint32_t)kGrtMaxDepth - 2u);
    if (todo && N > 2) hipMemsetAsync(todo, 0, 4, s);
    for (uint32_t p = 0; p < passes; ++p)
        hipLaunchKernelGGL(grt_refit_pass_kernel, dim3(div_up(N > 1 ? N - 1 : 1u, 256)), dim3(256), 0, s, N, p, aabb, slack, nodes, done,
                           (todo && N > 2 && p + 1 == passes) ? todo : nullptr);
    if (todo && N > 2) hipLaunchKernelGGL(grt_refit_finish_kernel, dim3(1), dim3(1024), 0, s, N, passes, aabb, slack, nodes, done, todo);
}

// (GRT_ONLY_DEGREE_4: development builds of kernel variants - scripts/build_variant.sh - instantiate the default kernel degree only: a
// quarter of the file's four minutes of compile time)
#ifdef GRT_ONLY_DEGREE_4
#define GRT_DISPATCH_DEGREE(DEG, ...) { constexpr int D_ = 4; __VA_ARGS__; }
#else
#define GRT_DISPATCH_DEGREE(DEG, ...)                          \
    switch (DEG) {                                             \
    case 0: { constexpr int D_ = 0; __VA_ARGS__; } break;      \
    case 1: { constexpr int D_ = 1; __VA_ARGS__; } break;      \
    case 2: { constexpr int D_ = 2; __VA_ARGS__; } break;      \
    case 3: { constexpr int D_ = 3; __VA_ARGS__; } break;      \
    case 5: { constexpr int D_ = 5; __VA_ARGS__; } break;      \
    case 8: { constexpr int D_ = 8; __VA_ARGS__; } break;      \
    default: { constexpr int D_ = 4; __VA_ARGS__; } break;     \
    }
#endif

// ---------------------------------------------------------------------------------------------
// packet lists: bounding cones of the ray packets, particle binning (the 3DGUT pipeline over cones instead of screen tiles)
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t blocks_x(int W) { return (uint32_t)(W + 7) / 8u; }
__host__ __device__ __forceinline__ uint32_t blocks_y(int H) { return (uint32_t)(H + 7) / 8u; }
uint32_t grt_num_blocks(int W, int H) { return blocks_x(W) * blocks_y(H); }
size_t grt_pair_cache_bytes(uint32_t N) { return (size_t)N * (4 * 16 + 4); }   // kBinCachedPairs uint4 + the pair count, per particle
size_t grt_cone_table_bytes(int W, int H) {
    const size_t nb = grt_num_blocks(W, H);
    return nb * (sizeof(GrtCone) + sizeof(GrtPyramid) + sizeof(float4)) + 64 + (size_t)(blocks_x(W) + blocks_y(H)) * sizeof(float2);
}
uint32_t grt_num_super(int W, int H) { return ((blocks_x(W) + 7u) / 8u) * ((blocks_y(H) + 7u) / 8u); }

// the entry offsets come from a 32-bit inclusive scan: a total beyond 2^32 shows as a descent somewhere in them
__global__ __launch_bounds__(256) void grt_list_check_kernel(uint32_t n, const uint32_t* __restrict__ offsets, uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > 0 && i < n && offsets[i] < offsets[i - 1]) flag[1] = 1u;
}
void grt_launch_list_check(hipStream_t s, uint32_t n, const uint32_t* offsets, uint32_t* flag) {
    hipLaunchKernelGGL(grt_list_check_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, n, offsets, flag);
}
__global__ void grt_list_init_kernel(uint32_t* __restrict__ flag, uint32_t* __restrict__ dir_len_enc, float* __restrict__ grid_hdr, uint32_t grid_ok) {
    grid_hdr[9] = __uint_as_float(grid_ok);   // cleared by a packet with a ray too far off the centre direction (grt_block_cone_kernel)
    flag[0] = 1u;
    flag[1] = 0u;   // set when the 32-bit entry count wrapped (grt_list_check_kernel)
    dir_len_enc[0] = 0x7F7FFFFFu;   // min |d| (bit patterns of positive floats order like integers)
    dir_len_enc[1] = 0u;            // max |d|
}
// one wave per 8x8 packet: bounding cone of its rays' directions, |d| range, and whether every ray starts where ray 0 does
__global__ __launch_bounds__(64) void grt_block_cone_kernel(GrtTraceParams P, const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                            uint32_t* __restrict__ flag, GrtCone* __restrict__ cones) {
    const uint32_t b = blockIdx.x, gx = blocks_x(P.W);
    const int lane = threadIdx.x;
    const int px = (int)(b % gx) * 8 + (lane & 7), py = (int)(b / gx) * 8 + (lane >> 3);
    const bool in_image = (px < P.W) && (py < P.H);
    const size_t pix = in_image ? (size_t)py * P.W + px : 0;
    const bool same = (__float_as_uint(ray_o[3 * pix]) == __float_as_uint(ray_o[0])) && (__float_as_uint(ray_o[3 * pix + 1]) == __float_as_uint(ray_o[1])) &&
                      (__float_as_uint(ray_o[3 * pix + 2]) == __float_as_uint(ray_o[2]));
    const RayW r = make_ray(P, ray_o, ray_d, pix);
    const float len = sqrtf(dot(r.d, r.d));
    const bool good = in_image && len > 0.f && len < 3.0e38f;
    const f3 dh = good ? r.d * (1.f / len) : mk3(0.f, 0.f, 0.f);
    const f3 sum = mk3(wave_sum(dh.x), wave_sum(dh.y), wave_sum(dh.z));
    const float sl = sqrtf(dot(sum, sum));
    const f3 axis = sl > 0.f ? sum * (1.f / sl) : mk3(0.f, 0.f, 1.f);
    const float cmin = wave_min(good ? dot(dh, axis) : 1.f);
    const float lmin = wave_min(good ? len : 3.0e38f), lmax = wave_max(good ? len : 0.f);
    const bool bad = __any(in_image && !good);   // a degenerate direction: no cone, no lists for this frame
    if (lane == 0) {
        GrtCone c;
        c.ax = axis.x; c.ay = axis.y; c.az = axis.z;
        // widened by rounding margins; a cone of 90 degrees or more (or no axis) holds everything
        const float ct = cmin * (1.f - 4e-6f) - 4e-6f;
        const bool all = !(ct > 0.05f) || !(sl > 0.f);
        c.cos_t = all ? -1.f : ct;
        c.sin_t = all ? 0.f : sqrtf(fmaxf(0.f, 1.f - ct * ct));
        c.valid = __any(in_image) ? 1.f : 0.f;
        c.pad0 = lmin; c.pad1 = lmax;   // |d| range of the packet: reduced per super tile (10 k atomics on two words cost 0.2 ms here)
        cones[b] = c;
        if (bad) flag[0] = 0u;
    }
    if (__any(in_image && !same) && lane == 0) flag[0] = 0u;
    // the packet's bounding pyramid (GrtPyramid): u along the pixel rows (the x-weighted mean of the directions, made perpendicular to
    // the axis), w = axis x u; the tangents of every ray against both, widened by rounding margins
    const float wx = good ? (float)(lane & 7) - 3.5f : 0.f;
    f3 u = mk3(wave_sum(wx * dh.x), wave_sum(wx * dh.y), wave_sum(wx * dh.z));
    u = u - axis * dot(u, axis);
    float ul = sqrtf(dot(u, u));
    if (!(ul > 1e-12f)) {   // (a one-column packet, or parallel rays: any perpendicular will do)
        u = fabsf(axis.x) < 0.6f ? mk3(1.f, 0.f, 0.f) : mk3(0.f, 1.f, 0.f);
        u = u - axis * dot(u, axis);
        ul = sqrtf(dot(u, u));
    }
    u = u * (1.f / ul);
    const f3 w = cross(axis, u);
    const float dz = dot(dh, axis);
    const bool fits = good && dz > 0.05f;
    const float tx = fits ? dot(dh, u) / dz : 0.f, ty = fits ? dot(dh, w) / dz : 0.f;
    const float x0 = wave_min(good ? tx : 3.0e38f), x1 = wave_max(good ? tx : -3.0e38f);
    const float y0 = wave_min(good ? ty : 3.0e38f), y1 = wave_max(good ? ty : -3.0e38f);
    const bool pyr_ok = !__any(good && !fits) && __any(good) && sl > 0.f;
    if (lane == 0) {
        GrtPyramid py;
        py.ux = u.x; py.uy = u.y; py.uz = u.z; py.wx = w.x; py.wy = w.y; py.wz = w.z;
        const float mx = 1e-5f * (1.f + fmaxf(fabsf(x0), fabsf(x1))), my = 1e-5f * (1.f + fmaxf(fabsf(y0), fabsf(y1)));
        py.x0 = x0 - mx; py.x1 = x1 + mx; py.y0 = y0 - my; py.y1 = y1 + my;
        py.ok = pyr_ok ? 1.f : 0.f;
        py.pad = 0.f;
        const_cast<GrtPyramid*>(grt_block_pyramids(cones, gridDim.x))[b] = py;
    }
    // the packet in the frame's tangent plane (GrtGrid): a = the centre pixel's direction, u = towards its neighbour in the row (every
    // wave derives the same frame from the same two rays)
    const GrtGrid G = grt_block_grid(cones, gridDim.x, gx);
    const int cpx = P.W / 2, cpy = P.H / 2, npx = cpx + 1 < P.W ? cpx + 1 : (cpx > 0 ? cpx - 1 : cpx);
    const RayW rc = make_ray(P, ray_o, ray_d, (size_t)cpy * P.W + cpx), rn = make_ray(P, ray_o, ray_d, (size_t)cpy * P.W + npx);
    const float lc = sqrtf(dot(rc.d, rc.d)), ln = sqrtf(dot(rn.d, rn.d));
    const bool frame_ok = lc > 0.f && lc < 3.0e38f;
    const f3 ag = frame_ok ? rc.d * (1.f / lc) : mk3(0.f, 0.f, 1.f);
    f3 ug = (ln > 0.f && ln < 3.0e38f) ? (rn.d * (1.f / ln) - ag) * (npx > cpx ? 1.f : -1.f) : mk3(0.f, 0.f, 0.f);
    ug = ug - ag * dot(ug, ag);
    float ugl = sqrtf(dot(ug, ug));
    if (!(ugl > 1e-12f)) {
        ug = fabsf(ag.x) < 0.6f ? mk3(1.f, 0.f, 0.f) : mk3(0.f, 1.f, 0.f);
        ug = ug - ag * dot(ug, ag);
        ugl = sqrtf(dot(ug, ug));
    }
    ug = ug * (1.f / ugl);
    const f3 wg = cross(ag, ug);
    const float gz = dot(dh, ag);
    const bool gfits = good && gz > 0.05f;
    const float gtx = gfits ? dot(dh, ug) / gz : 0.f, gty = gfits ? dot(dh, wg) / gz : 0.f;
    const float gx0 = wave_min(gfits ? gtx : 3.0e38f), gx1 = wave_max(gfits ? gtx : -3.0e38f);
    const float gy0 = wave_min(gfits ? gty : 3.0e38f), gy1 = wave_max(gfits ? gty : -3.0e38f);
    const bool grid_bad = __any(good && !gfits) || !frame_ok;
    if (lane == 0) {
        const float mx = 2e-5f * (1.f + fmaxf(fabsf(gx0), fabsf(gx1))), my = 2e-5f * (1.f + fmaxf(fabsf(gy0), fabsf(gy1)));
        G.rects[b] = make_float4(gx0 - mx, gx1 + mx, gy0 - my, gy1 + my);   // (a packet without a usable ray: an empty rectangle)
        if (grid_bad) G.hdr[9] = __uint_as_float(0u);
        if (b == 0) {
            G.hdr[0] = ag.x; G.hdr[1] = ag.y; G.hdr[2] = ag.z; G.hdr[3] = ug.x; G.hdr[4] = ug.y; G.hdr[5] = ug.z;
            G.hdr[6] = wg.x; G.hdr[7] = wg.y; G.hdr[8] = wg.z;
        }
    }
}
// the union of the tangent-plane rectangles over each packet column and each packet row
// (one wave per column / row, the lanes stride over its packets: a thread per column walked its ~100 rectangles as a chain of dependent
// loads - 48 us on the forward's critical path for 200 intervals)
__global__ __launch_bounds__(64) void grt_grid_tables_kernel(GrtTraceParams P, GrtCone* __restrict__ cones) {
    const uint32_t gx = blocks_x(P.W), gy = blocks_y(P.H), i = blockIdx.x;
    const int lane = threadIdx.x;
    const GrtGrid G = grt_block_grid(cones, gx * gy, gx);
    if (i >= gx + gy) return;
    float lo = 3.0e38f, hi = -3.0e38f;
    if (i < gx) {
        for (uint32_t r = (uint32_t)lane; r < gy; r += 64u) { const float4 q = G.rects[r * gx + i]; lo = fminf(lo, q.x); hi = fmaxf(hi, q.y); }
    } else {
        for (uint32_t c = (uint32_t)lane; c < gx; c += 64u) { const float4 q = G.rects[(i - gx) * gx + c]; lo = fminf(lo, q.z); hi = fmaxf(hi, q.w); }
    }
    lo = wave_extreme<false>(lo); hi = wave_extreme<true>(hi);
    if (lane == 0) {
        if (i < gx) G.cols[i] = make_float2(lo, hi);
        else G.rows[i - gx] = make_float2(lo, hi);
    }
}
// one wave per super tile (8x8 packets): cone around its packets' cones
__global__ __launch_bounds__(64) void grt_super_cone_kernel(GrtTraceParams P, const GrtCone* __restrict__ cones, GrtCone* __restrict__ super_cones,
                                                            uint32_t* __restrict__ dir_len_enc) {
    const uint32_t s = blockIdx.x, gx = blocks_x(P.W), gy = blocks_y(P.H), sx = (gx + 7u) / 8u;
    const int lane = threadIdx.x;
    const uint32_t bx = (s % sx) * 8u + (uint32_t)(lane & 7), by = (s / sx) * 8u + (uint32_t)(lane >> 3);
    const bool have = bx < gx && by < gy;
    GrtCone c = {0.f, 0.f, 1.f, 1.f, 0.f, 0.f, 0.f, 0.f};
    if (have) c = cones[by * gx + bx];
    const bool use = have && c.valid != 0.f;
    const f3 a = use ? mk3(c.ax, c.ay, c.az) : mk3(0.f, 0.f, 0.f);
    const f3 sum = mk3(wave_sum(a.x), wave_sum(a.y), wave_sum(a.z));
    const float sl = sqrtf(dot(sum, sum));
    const f3 axis = sl > 0.f ? sum * (1.f / sl) : mk3(0.f, 0.f, 1.f);
    float ang = 0.f;
    if (use) ang = (c.cos_t <= -1.f) ? 4.f : acosf(fminf(1.f, fmaxf(-1.f, dot(a, axis)))) + acosf(fminf(1.f, c.cos_t)) + 1e-5f;
    const float amax = wave_max(ang);
    const float lmin = wave_min(use ? c.pad0 : 3.0e38f), lmax = wave_max(use ? c.pad1 : 0.f);
    if (lane == 0) {
        if (lmin < 3.0e38f) {
            atomicMin(&dir_len_enc[0], __float_as_uint(lmin));
            atomicMax(&dir_len_enc[1], __float_as_uint(lmax));
        }
        GrtCone o;
        o.ax = axis.x; o.ay = axis.y; o.az = axis.z;
        const bool all = !(amax < 1.5f) || !(sl > 0.f);
        o.cos_t = all ? -1.f : cosf(amax);
        o.sin_t = all ? 0.f : sinf(amax);
        o.valid = __any(use) ? 1.f : 0.f;
        o.pad0 = o.pad1 = 0.f;
        super_cones[s] = o;
    }
}
// does the sphere (centre v relative to the apex, |v|^2 = L2, radius R) reach into the cone?  distance from the centre to the cone's
// surface <= s cos - c sin with c, s the centre's coordinates along / across the axis (a projection: never above the true distance)
__device__ __forceinline__ bool cone_hit(const GrtCone& k, f3 v, float L2, float R) {
    const float c = v.x * k.ax + v.y * k.ay + v.z * k.az;
    const float sq = sqrtf(fmaxf(0.f, L2 - c * c));
    return (k.valid != 0.f) && (sq * k.cos_t - c * k.sin_t <= R * 1.00001f + 1e-12f);
}
struct BinParticle {
    f3 v;            // proxy centre relative to the ray origin
    float L2, Rs;    // |v|^2, radius of the proxy box's bounding sphere
    float key, ub;   // bounds of the hit distance t for any ray of the frame
    f3 h0, h1, h2;   // the proxy box's half axes in world space: box = centre + s0 h0 + s1 h1 + s2 h2, |s| <= 1
};
// Does the proxy BOX reach into the packet?  Every ray of the packet lies in the cone and in the pyramid, both convex with the apex at
// the common origin; a plane through the apex with the whole packet on one side and the whole box strictly on the other separates
// them, and a separated box cannot be hit by any ray of the packet.  Tested: the cone's tangent plane facing the box centre (normal
// n = cos r - sin a, r the unit vector from the axis to the centre: n.v is the sphere test's distance, here compared with the box's
// half width along n instead of the sphere's radius), the pyramid's four side planes and the plane behind the apex.  Each comparison
// carries a margin above the rounding of its two sides (relative 1e-5 of |v| + R, i.e. about 1/100 of a pixel).
__device__ __forceinline__ bool packet_hit(const GrtCone& k, const GrtPyramid& py, f3 v, float L2, float Rs, f3 h0, f3 h1, f3 h2, bool sphere_only) {
    const f3 a = mk3(k.ax, k.ay, k.az);
    const float c = dot(v, a);
    const float sq = sqrtf(fmaxf(0.f, L2 - c * c));
    const float s = sq * k.cos_t - c * k.sin_t;
    if (k.valid == 0.f || !(s <= Rs * 1.00001f + 1e-12f)) return false;
    if (sphere_only || k.cos_t <= -1.f) return true;
    const float L = sqrtf(L2), margin = 1e-5f * (L + Rs);
    const float ha0 = dot(h0, a), ha1 = dot(h1, a), ha2 = dot(h2, a);
    if (sq > 1e-4f * L) {
        const float ic = k.cos_t / sq;
        // n.h = cos (h.v - c h.a) / sq - sin h.a
        const float w0 = (dot(h0, v) - c * ha0) * ic - k.sin_t * ha0, w1 = (dot(h1, v) - c * ha1) * ic - k.sin_t * ha1,
                    w2 = (dot(h2, v) - c * ha2) * ic - k.sin_t * ha2;
        if (s > (fabsf(w0) + fabsf(w1) + fabsf(w2)) * 1.00001f + margin) return false;
    }
    if (-c > (fabsf(ha0) + fabsf(ha1) + fabsf(ha2)) * 1.00001f + margin) return false;   // wholly behind the apex
    if (py.ok != 0.f) {
        const f3 u = mk3(py.ux, py.uy, py.uz), w = mk3(py.wx, py.wy, py.wz);
        const float vu = dot(v, u), vw = dot(v, w);
        const float hu0 = dot(h0, u), hu1 = dot(h1, u), hu2 = dot(h2, u), hw0 = dot(h0, w), hw1 = dot(h1, w), hw2 = dot(h2, w);
        const float m2 = margin * 2.f;   // (|u - x a| <= sqrt(1 + x^2) < 2 for tangents below 1.7)
        // plane u - x1 a: the packet has n.p <= 0
        if (vu - py.x1 * c > (fabsf(hu0 - py.x1 * ha0) + fabsf(hu1 - py.x1 * ha1) + fabsf(hu2 - py.x1 * ha2)) * 1.00001f + m2 * (1.f + fabsf(py.x1))) return false;
        if (py.x0 * c - vu > (fabsf(hu0 - py.x0 * ha0) + fabsf(hu1 - py.x0 * ha1) + fabsf(hu2 - py.x0 * ha2)) * 1.00001f + m2 * (1.f + fabsf(py.x0))) return false;
        if (vw - py.y1 * c > (fabsf(hw0 - py.y1 * ha0) + fabsf(hw1 - py.y1 * ha1) + fabsf(hw2 - py.y1 * ha2)) * 1.00001f + m2 * (1.f + fabsf(py.y1))) return false;
        if (py.y0 * c - vw > (fabsf(hw0 - py.y0 * ha0) + fabsf(hw1 - py.y0 * ha1) + fabsf(hw2 - py.y0 * ha2)) * 1.00001f + m2 * (1.f + fabsf(py.y0))) return false;
    }
    return true;
}
// bounds: the hit "distance" t of a candidate is the ray parameter of the point closest to the centre in the proxy's metric; that point
// lies within sqrt(3) max(kscl) of the centre whenever the ray touches the proxy box (the box holds a point of the ray at metric distance
// <= sqrt 3 and the closest one is no farther), so |o + t d - mu| <= Rt and (|v| - Rt) / |d| <= t <= (|v| + Rt) / |d|
__device__ __forceinline__ BinParticle bin_particle(const float4& a, const float4& b, const float4& e, f3 o, float dmin, float dmax, int prim, uint32_t id,
                                                    const float* __restrict__ box8 = nullptr) {
    BinParticle q;
    float k0 = 1.f / sqrtf(a.x * a.x + a.y * a.y + a.z * a.z), k1 = 1.f / sqrtf(a.w * a.w + b.x * b.x + b.y * b.y),
          k2 = 1.f / sqrtf(b.z * b.z + b.w * b.w + e.x * e.x);   // rows of W = R^T / kscl
    // half axis i = (row i of W) / |row i|^2: W h_i = e_i, the box is |W (x - mu)|_inf <= 1
    q.h0 = mk3(a.x, a.y, a.z) * (k0 * k0); q.h1 = mk3(a.w, b.x, b.y) * (k1 * k1); q.h2 = mk3(b.z, b.w, e.x) * (k2 * k2);
    float Rt = 1.7320509f * fmaxf(k0, fmaxf(k1, k2)) * 1.00001f;
    if (prim == GRUT_PRIM_CUSTOM) {
        // custom primitives (round 6): the intersection program runs for the rays that cross the particle's WORLD box (box8, the reference's
        // AABB kernel), so THAT box - axis-aligned half vectors, a hair padded over the rounding of its centre - stands in every separation
        // test; a hit that can be accepted lies within 3 sigma of the scale frame, |W (x* - mu)| ks < 3, i.e. within 3 kmax / ks of the centre
        const float* bx = box8 + 8 * (size_t)id;
        const float ux = 0.5f * (bx[3] - bx[0]) * 1.00001f + 1e-6f * (fabsf(e.y) + 1.f), uy = 0.5f * (bx[4] - bx[1]) * 1.00001f + 1e-6f * (fabsf(e.z) + 1.f),
                    uz = 0.5f * (bx[5] - bx[2]) * 1.00001f + 1e-6f * (fabsf(e.w) + 1.f);
        Rt = 3.f * fmaxf(k0, fmaxf(k1, k2)) / sqrtf(bx[6]) * 1.0001f;
        q.h0 = mk3(ux, 0.f, 0.f); q.h1 = mk3(0.f, uy, 0.f); q.h2 = mk3(0.f, 0.f, uz);
        k0 = ux; k1 = uy; k2 = uz;
    } else if (prim != GRUT_PRIM_INSTANCES) {
        // triangle-mesh proxies: the box of the polyhedron's vertices (half extents ext_i along the proxy's axes) stands in for the unit
        // cube in every separation test, and the reported distance is the ENTRY into the polyhedron - a point of that box: within its
        // bounding sphere of the centre's distance along any ray
        float ext[3];
        proxy_extents(prim, id, ext);
        k0 *= ext[0]; k1 *= ext[1]; k2 *= ext[2];
        q.h0 = q.h0 * ext[0]; q.h1 = q.h1 * ext[1]; q.h2 = q.h2 * ext[2];
        Rt = sqrtf(k0 * k0 + k1 * k1 + k2 * k2) * 1.00002f;
    }
    q.Rs = sqrtf(k0 * k0 + k1 * k1 + k2 * k2) * 1.00001f;
    q.v = mk3(e.y - o.x, e.z - o.y, e.w - o.z);
    q.L2 = dot(q.v, q.v);
    const float L = sqrtf(q.L2);
    const float lo = L * (1.f - 2e-6f) - Rt;
    q.key = lo > 0.f ? (lo / dmax) * (1.f - 2e-6f) : 0.f;           // (a hit needs t > 0)
    // the key only has to be A lower bound that the lists ascend in: truncated to its upper 16 bits (sign, exponent, 7 mantissa bits: at most
    // 0.8 % below) the particles sort in two radix passes instead of four; the hit order is made of the rays' own distances, not of keys
    q.key = __uint_as_float(__float_as_uint(q.key) & 0xFFFF0000u);
    q.ub = ((L * (1.f + 2e-6f) + Rt) / dmin) * (1.f + 2e-6f) + 1e-30f;
    return q;
}
// The packets a particle's bounding sphere reaches, for the 64 particles of a wave: the super tiles are visited by all lanes together
// (each lane tests its own particle against the super tile's cone); for every (particle, super tile) pair that passes, the 64 lanes
// test the super tile's 64 packets for that one particle — a particle that covers the screen costs the wave one step per super tile,
// not one lane 64 steps per super tile.  EMIT: write the entries (counting pass otherwise).
struct BinOut {
    uint32_t *block_keys, *vals;   // sort key = packet, payload = particle (the sorted payloads ARE the lists)
    // The counting pass leaves what it found — per particle up to kBinCachedPairs (super tile, 64-bit packet mask) pairs and their number —
    // so that the emitting pass writes the entries from the masks instead of testing every packet again (1.28 -> ms; a particle with
    // more pairs than the cache holds is tested again)
    uint4* pairs;       // [N][kBinCachedPairs] {super tile, mask lo, mask hi, -}
    uint32_t* pair_n;   // [N] pairs found (may exceed the cache)
};
constexpr int kBinCachedPairs = 4;
template <bool EMIT>
__device__ __forceinline__ uint32_t bin_pairs(const GrtTraceParams& P, const GrtCone* __restrict__ block_cones, const GrtCone* __restrict__ super_cones,
                                              int lane, bool have, const BinParticle& q, uint32_t pid, uint32_t off, uint32_t end, const BinOut& out) {
    const uint32_t gx = blocks_x(P.W), gy = blocks_y(P.H), sx = (gx + 7u) / 8u, sy = (gy + 7u) / 8u;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    uint32_t n = 0, np = 0;
    auto bc = [](float x, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), src)); };
    for (uint32_t s = 0; s < sx * sy; ++s) {
        // (the super tile's cone against the box too: the bounding sphere of a needle reaches super tiles its box stays clear of)
        const GrtPyramid no_pyramid = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        unsigned long long m = __ballot(have && packet_hit(super_cones[s], no_pyramid, q.v, q.L2, q.Rs, q.h0, q.h1, q.h2, P.sphere_lists != 0));
        const uint32_t bx = (s % sx) * 8u + (uint32_t)(lane & 7), by = (s / sx) * 8u + (uint32_t)(lane >> 3);
        const bool exists = bx < gx && by < gy;
        const uint32_t b = exists ? by * gx + bx : 0u;
        GrtCone kc = {0.f, 0.f, 1.f, 1.f, 0.f, 0.f, 0.f, 0.f};
        GrtPyramid kp = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (m && exists) {
            kc = block_cones[b];
            kp = grt_block_pyramids(block_cones, gx * gy)[b];
        }
        while (m) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            const f3 v = mk3(bc(q.v.x, src), bc(q.v.y, src), bc(q.v.z, src));
            const float L2 = bc(q.L2, src), Rs = bc(q.Rs, src);
            const f3 h0 = mk3(bc(q.h0.x, src), bc(q.h0.y, src), bc(q.h0.z, src)), h1 = mk3(bc(q.h1.x, src), bc(q.h1.y, src), bc(q.h1.z, src)),
                     h2 = mk3(bc(q.h2.x, src), bc(q.h2.y, src), bc(q.h2.z, src));
            const bool hit = exists && packet_hit(kc, kp, v, L2, Rs, h0, h1, h2, P.sphere_lists != 0);
            const unsigned long long hm = __ballot(hit);
            const uint32_t cnt = (uint32_t)__popcll(hm);
            if (EMIT) {
                const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)off, src), e = (uint32_t)__builtin_amdgcn_readlane((int)end, src);
                const uint32_t slot = o + (uint32_t)__popcll(hm & lt);
                if (hit && slot < e) {
                    out.block_keys[slot] = b;
                    out.vals[slot] = (uint32_t)__builtin_amdgcn_readlane((int)pid, src);
                }
                if (lane == src) off += cnt;
            } else if (lane == src) {
                n += cnt;
                if (cnt) {
                    if (np < (uint32_t)kBinCachedPairs) out.pairs[(size_t)pid * kBinCachedPairs + np] = make_uint4(s, (uint32_t)hm, (uint32_t)(hm >> 32), 0u);
                    np++;
                }
            }
        }
    }
    if (EMIT) {
        for (; off < end; ++off) {   // (same tests as the counting pass: not expected)
            out.block_keys[off] = 0xFFFFFFFFu; out.vals[off] = 0xFFFFFFFFu;
        }
    } else if (have) {
        out.pair_n[pid] = np;
    }
    return n;
}
// the emitting pass for the particles whose pairs the counting pass cached: pair by pair, the wave writes the packets of one mask
__device__ __forceinline__ void emit_cached_pairs(const GrtTraceParams& P, int lane, bool cached, uint32_t pid, uint32_t np, uint32_t& off, uint32_t end,
                                                  const BinOut& out) {
    const uint32_t gx = blocks_x(P.W), gy = blocks_y(P.H), sx = (gx + 7u) / 8u;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll 1
    for (int k = 0; k < kBinCachedPairs; ++k) {
        unsigned long long m = __ballot(cached && (uint32_t)k < np);
        if (!m) break;
        uint4 pr = make_uint4(0u, 0u, 0u, 0u);
        if (cached && (uint32_t)k < np) pr = out.pairs[(size_t)pid * kBinCachedPairs + k];
        while (m) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            const uint32_t s = (uint32_t)__builtin_amdgcn_readlane((int)pr.x, src);
            const unsigned long long hm = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)pr.y, src) |
                                          ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)pr.z, src) << 32);
            const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)off, src), e = (uint32_t)__builtin_amdgcn_readlane((int)end, src);
            const uint32_t bx = (s % sx) * 8u + (uint32_t)(lane & 7), by = (s / sx) * 8u + (uint32_t)(lane >> 3);
            const uint32_t slot = o + (uint32_t)__popcll(hm & lt);
            if (((hm >> lane) & 1ull) && slot < e) {
                out.block_keys[slot] = by * gx + bx;
                out.vals[slot] = (uint32_t)__builtin_amdgcn_readlane((int)pid, src);
            }
            if (lane == src) off += (uint32_t)__popcll(hm);
        }
    }
}
// ---- binning over the frame's tangent plane (GrtGrid) ---------------------------------------------------------------------------
// The particle's candidate packets: the 8 corners of its proxy box projected onto the plane give a rectangle [px0, px1] x [py0, py1]
// that holds (d.u)/(d.a), (d.w)/(d.a) of every ray through the box; only packets whose own rectangle overlaps it can hold such a ray.
//   kind 0  no packet (the box lies behind the apex plane: every ray of the frame has d.a > 0)
//   kind 1  the packets of at most kBinLaneArea (column, row) cells: the lane tests them itself
//   kind 2  more cells: the wave tests them together, 64 at a time
//   kind 3  no rectangle — the box comes within 2 % of its distance of the apex plane, where the projection blows up: the wave walks
//           the super tiles for it (super tile cones, then the packets of those it reaches)
// The rectangle only SELECTS candidates; what enters a list is decided by packet_hit, as in the super tile scan.
constexpr uint32_t kBinLaneArea = 32;   // (GRUT_GRT_LANE_AREA: 8 / 16 / 32 / 64 cells -> count + expand 1.07 / 0.85 / 0.72 / 0.77 ms at 1 M particles, 800x800)
struct GridParticle {
    int kind;
    uint32_t bx0, by0, wdt, hgt;
    float px0, px1, py0, py1;
};
__device__ __forceinline__ GridParticle grid_particle(const GrtGrid& G, uint32_t gx, uint32_t gy, bool have, const BinParticle& q, uint32_t lane_area = kBinLaneArea) {
    GridParticle g;
    g.kind = 0; g.bx0 = g.by0 = 0u; g.wdt = g.hgt = 1u; g.px0 = g.py0 = 0.f; g.px1 = g.py1 = 0.f;
    const f3 a = mk3(G.hdr[0], G.hdr[1], G.hdr[2]), u = mk3(G.hdr[3], G.hdr[4], G.hdr[5]), w = mk3(G.hdr[6], G.hdr[7], G.hdr[8]);
    const float va = dot(q.v, a), vu = dot(q.v, u), vw = dot(q.v, w);
    const float a0 = dot(q.h0, a), a1 = dot(q.h1, a), a2 = dot(q.h2, a), u0 = dot(q.h0, u), u1 = dot(q.h1, u), u2 = dot(q.h2, u),
                w0 = dot(q.h0, w), w1 = dot(q.h1, w), w2 = dot(q.h2, w);
    float zmin = 3.0e38f, zmax = -3.0e38f, x0 = 3.0e38f, x1 = -3.0e38f, y0 = 3.0e38f, y1 = -3.0e38f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float s0 = (c & 1) ? 1.f : -1.f, s1 = (c & 2) ? 1.f : -1.f, s2 = (c & 4) ? 1.f : -1.f;
        const float z = va + s0 * a0 + s1 * a1 + s2 * a2, xx = vu + s0 * u0 + s1 * u1 + s2 * u2, yy = vw + s0 * w0 + s1 * w1 + s2 * w2;
        zmin = fminf(zmin, z); zmax = fmaxf(zmax, z);
        const float iz = __builtin_amdgcn_rcpf(fmaxf(z, 1e-30f));
        x0 = fminf(x0, xx * iz); x1 = fmaxf(x1, xx * iz); y0 = fminf(y0, yy * iz); y1 = fmaxf(y1, yy * iz);
    }
    const float ext = sqrtf(q.L2) + q.Rs;
    if (!have || !(zmax > -1e-5f * ext)) return g;
    // (rounding: a corner's z carries ~3e-7 ext, so with z >= 0.02 ext the quotients are good to 2e-5 (1 + |x|); NaNs take the walk)
    const float mx = 1e-4f * (1.f + fmaxf(fabsf(x0), fabsf(x1))), my = 1e-4f * (1.f + fmaxf(fabsf(y0), fabsf(y1)));
    g.px0 = x0 - mx; g.px1 = x1 + mx; g.py0 = y0 - my; g.py1 = y1 + my;
    if (!(zmin >= 0.02f * ext) || !(g.px0 <= g.px1) || !(g.py0 <= g.py1)) { g.kind = 3; return g; }
    uint32_t bx0 = gx, bx1 = 0u, by0 = gy, by1 = 0u;
    for (uint32_t c = 0; c < gx; ++c) {
        const float2 iv = G.cols[c];
        if (iv.x <= g.px1 && iv.y >= g.px0) { bx0 = min(bx0, c); bx1 = c; }
    }
    for (uint32_t r = 0; r < gy; ++r) {
        const float2 iv = G.rows[r];
        if (iv.x <= g.py1 && iv.y >= g.py0) { by0 = min(by0, r); by1 = r; }
    }
    if (bx0 >= gx || by0 >= gy) return g;   // outside the frame
    g.bx0 = bx0; g.by0 = by0; g.wdt = bx1 - bx0 + 1u; g.hgt = by1 - by0 + 1u;
    g.kind = (g.wdt * g.hgt <= lane_area) ? 1 : 2;
    return g;
}
__device__ __forceinline__ bool rect_overlap(const float4& rc, float px0, float px1, float py0, float py1) {
    return rc.x <= px1 && rc.y >= px0 && rc.z <= py1 && rc.w >= py0;
}
// kind 1, counting pass: the lane's own cells, first against the packets' rectangles, then the survivors against packet_hit;
// returns the mask of the cells (row-major in the particle's rectangle) whose packets the box reaches
__device__ __forceinline__ unsigned long long grid_lane_cells(const GrtGrid& G, const GrtCone* __restrict__ block_cones, uint32_t gx, uint32_t gy,
                                                              const GridParticle& g, const BinParticle& q) {
    const bool mine = g.kind == 1;
    const uint32_t area = mine ? g.wdt * g.hgt : 0u;
    unsigned long long cand = 0ull;
    uint32_t cx = 0u, cy = 0u;
    for (uint32_t j = 0; __any(j < area); ++j) {
        if (j < area) {
            const float4 rc = G.rects[(g.by0 + cy) * gx + g.bx0 + cx];
            if (rect_overlap(rc, g.px0, g.px1, g.py0, g.py1)) cand |= 1ull << j;
            if (++cx == g.wdt) { cx = 0u; ++cy; }
        }
    }
    const GrtPyramid* pyramids = grt_block_pyramids(block_cones, gx * gy);
    const float iw = 1.f / (float)g.wdt;
    unsigned long long hm = 0ull;
    while (__any(cand != 0ull)) {
        const bool act = cand != 0ull;
        const int j = act ? __ffsll((long long)cand) - 1 : 0;
        cand &= cand - 1ull;   // (0 stays 0)
        const uint32_t ry = (uint32_t)(((float)j + 0.5f) * iw), rx = (uint32_t)j - ry * g.wdt;   // exact: j < 64, the quotient stays 1/128 clear of an integer
        const uint32_t b = act ? (g.by0 + ry) * gx + g.bx0 + rx : 0u;
        const GrtCone kc = block_cones[b];
        const GrtPyramid kp = pyramids[b];
        if (act && packet_hit(kc, kp, q.v, q.L2, q.Rs, q.h0, q.h1, q.h2, false)) hm |= 1ull << j;
    }
    return hm;
}
// kinds 2 and 3: one particle at a time, the 64 lanes on 64 of its candidate packets.  EMIT: the entries are written at the
// particle's [off, end); the number of packets reached is added to lane src's n
template <bool EMIT>
__device__ __forceinline__ void grid_wave_cells(const GrtTraceParams& P, const GrtGrid& G, const GrtCone* __restrict__ block_cones,
                                                const GrtCone* __restrict__ super_cones, int lane, const GridParticle& g, const BinParticle& q,
                                                uint32_t pid, uint32_t& n, uint32_t& off, uint32_t end, const BinOut& out) {
    const uint32_t gx = blocks_x(P.W), gy = blocks_y(P.H), sx = (gx + 7u) / 8u, sy = (gy + 7u) / 8u;
    const GrtPyramid* pyramids = grt_block_pyramids(block_cones, gx * gy);
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    auto bc = [](float x, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), src)); };
    auto bu = [](uint32_t x, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)x, src); };
    unsigned long long m = __ballot(g.kind >= 2);
    while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const f3 v = mk3(bc(q.v.x, src), bc(q.v.y, src), bc(q.v.z, src));
        const float L2 = bc(q.L2, src), Rs = bc(q.Rs, src);
        const f3 h0 = mk3(bc(q.h0.x, src), bc(q.h0.y, src), bc(q.h0.z, src)), h1 = mk3(bc(q.h1.x, src), bc(q.h1.y, src), bc(q.h1.z, src)),
                 h2 = mk3(bc(q.h2.x, src), bc(q.h2.y, src), bc(q.h2.z, src));
        const uint32_t o = EMIT ? bu(off, src) : 0u, e = EMIT ? bu(end, src) : 0u, id = EMIT ? bu(pid, src) : 0u;
        uint32_t found = 0u;
        auto take = [&](bool hit, uint32_t b) {
            const unsigned long long hm = __ballot(hit);
            if (EMIT) {
                const uint32_t slot = o + found + (uint32_t)__popcll(hm & lt);
                if (hit && slot < e) { out.block_keys[slot] = b; out.vals[slot] = id; }
            }
            found += (uint32_t)__popcll(hm);
        };
        if (bu((uint32_t)g.kind, src) == 2u) {
            const uint32_t bx0 = bu(g.bx0, src), by0 = bu(g.by0, src), wdt = bu(g.wdt, src), total = wdt * bu(g.hgt, src);
            const float px0 = bc(g.px0, src), px1 = bc(g.px1, src), py0 = bc(g.py0, src), py1 = bc(g.py1, src);
            for (uint32_t c0 = 0; c0 < total; c0 += 64u) {
                const uint32_t c = c0 + (uint32_t)lane;
                const bool valid = c < total;
                const uint32_t cc = valid ? c : 0u, ry = cc / wdt, rx = cc - ry * wdt;
                const uint32_t b = (by0 + ry) * gx + bx0 + rx;
                bool hit = valid && rect_overlap(G.rects[b], px0, px1, py0, py1);
                if (__any(hit)) {
                    GrtCone kc = {0.f, 0.f, 1.f, 1.f, 0.f, 0.f, 0.f, 0.f};
                    GrtPyramid kp = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (hit) { kc = block_cones[b]; kp = pyramids[b]; }
                    hit = hit && packet_hit(kc, kp, v, L2, Rs, h0, h1, h2, false);
                }
                take(hit, b);
            }
        } else {
            const GrtPyramid no_pyramid = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (uint32_t s0 = 0; s0 < sx * sy; s0 += 64u) {
                const uint32_t sl = s0 + (uint32_t)lane;
                const bool reach = sl < sx * sy && packet_hit(super_cones[sl < sx * sy ? sl : 0u], no_pyramid, v, L2, Rs, h0, h1, h2, false);
                unsigned long long sm = __ballot(reach);
                while (sm) {
                    const uint32_t s = s0 + (uint32_t)(__ffsll((long long)sm) - 1);
                    sm &= sm - 1;
                    const uint32_t bx = (s % sx) * 8u + (uint32_t)(lane & 7), by = (s / sx) * 8u + (uint32_t)(lane >> 3);
                    const bool exists = bx < gx && by < gy;
                    const uint32_t b = exists ? by * gx + bx : 0u;
                    GrtCone kc = {0.f, 0.f, 1.f, 1.f, 0.f, 0.f, 0.f, 0.f};
                    GrtPyramid kp = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (exists) { kc = block_cones[b]; kp = pyramids[b]; }
                    take(exists && packet_hit(kc, kp, v, L2, Rs, h0, h1, h2, false), b);
                }
            }
        }
        if (lane == src) { n += found; if (EMIT) off += found; }
    }
}
// cache word of a kind-1 particle: its rectangle's corner and width (the mask of cells follows in the same uint4)
__device__ __forceinline__ uint32_t grid_pack(const GridParticle& g) { return g.bx0 | (g.by0 << 12) | ((g.wdt - 1u) << 24); }
constexpr uint32_t kGridWaveTested = 0xFFFFFFFFu;   // pair_n of a particle of kind 2 / 3: the emitting pass tests again

__global__ __launch_bounds__(256) void grt_list_count_kernel(GrtTraceParams P, GrtBvh bvh, const float* __restrict__ ray_o,
                                                             const uint32_t* __restrict__ flag, const uint32_t* __restrict__ dir_len_enc,
                                                             const GrtCone* __restrict__ block_cones, const GrtCone* __restrict__ super_cones,
                                                             float* __restrict__ inst_rel, uint32_t* __restrict__ key_bits,
                                                             uint32_t* __restrict__ counts, uint32_t* __restrict__ particle_idx, uint4* __restrict__ pairs,
                                                             uint32_t* __restrict__ pair_n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool have = i < bvh.N && flag[0] != 0u;
    float4 a = make_float4(1.f, 0.f, 0.f, 0.f), b = make_float4(1.f, 0.f, 0.f, 0.f), e = make_float4(1.f, 0.f, 0.f, 0.f);
    const f3 o = world_origin(P, mk3(ray_o[0], ray_o[1], ray_o[2]));
    const float dmin = __uint_as_float(dir_len_enc[0]), dmax = __uint_as_float(dir_len_enc[1]);
    if (have) {
        const float4* rec = reinterpret_cast<const float4*>(bvh.inst) + 3 * (size_t)i;
        a = rec[0]; b = rec[1]; e = rec[2];
    }
    const BinParticle q = bin_particle(a, b, e, o, dmin, dmax, P.prim, i, P.box8);
    const BinOut cache = {nullptr, nullptr, pairs, pair_n};
    const uint32_t gx = blocks_x(P.W), gy = blocks_y(P.H);
    const GrtGrid G = grt_block_grid(block_cones, gx * gy, gx);
    uint32_t n = 0u;
    if (__float_as_uint(G.hdr[9]) != 0u) {   // the frame has a tangent plane
        const GridParticle g = grid_particle(G, gx, gy, have, q, P.bin_lane_area);
        const unsigned long long hm = grid_lane_cells(G, block_cones, gx, gy, g, q);
        n = (uint32_t)__popcll(hm);
        uint32_t off = 0u;
        grid_wave_cells<false>(P, G, block_cones, super_cones, lane, g, q, i, n, off, 0u, cache);
        if (have) {
            pairs[(size_t)i * kBinCachedPairs] = make_uint4(grid_pack(g), (uint32_t)hm, (uint32_t)(hm >> 32), 0u);
            pair_n[i] = g.kind >= 2 ? kGridWaveTested : 0u;
        }
    } else {
        n = bin_pairs<false>(P, block_cones, super_cones, lane, have, q, i, 0u, 0u, cache);
    }
    if (i >= bvh.N) return;
    counts[i] = have ? n : 0u;   // (particle_idx, the payload of the key sort, is an iota: generated by the sort's first pass)
    key_bits[i] = (have && n) ? __float_as_uint(q.key) : 0xFFFFFFFFu;   // sort key (the sort consumes this array); particles no packet can reach go last
    if (!have) return;
    // the frame-relative record of the particle, one 64-byte line: {W rows, W (o - mu)} — the proxy-frame ray origin with the very
    // operations of the candidate test (candidate_abe<true> reads it) — and {proxy centre relative to the ray origin, sort key},
    // what list_round needs per entry for its packet-specific bounds
    const f3 po = proxy_origin(a, b, e, o);
    float4* out = reinterpret_cast<float4*>(inst_rel) + 4 * (size_t)i;
    out[0] = a; out[1] = b; out[2] = make_float4(e.x, po.x, po.y, po.z);
    out[3] = make_float4(q.v.x, q.v.y, q.v.z, q.key);
}
// where each particle's entries go: rank r of the key order owns [offsets[r-1], offsets[r])
__global__ __launch_bounds__(256) void grt_list_starts_kernel(uint32_t N, const uint32_t* __restrict__ rank_to_particle, const uint32_t* __restrict__ offsets,
                                                              uint32_t* __restrict__ starts) {
    const uint32_t rk = blockIdx.x * blockDim.x + threadIdx.x;
    if (rk < N) starts[rank_to_particle[rk]] = rk == 0 ? 0u : offsets[rk - 1];
}
// particle p writes its entries at [starts[p], starts[p] + counts[p]): (packet, particle) pairs.  The threads follow the PARTICLE
// order, not the key order: the nearest particles are the ones that cover hundreds of packets, and in key order they all sat in
// the first few waves (0.69 -> ms)
__global__ __launch_bounds__(256) void grt_list_expand_kernel(GrtTraceParams P, GrtBvh bvh, const float* __restrict__ ray_o,
                                                              const uint32_t* __restrict__ flag, const uint32_t* __restrict__ dir_len_enc,
                                                              const GrtCone* __restrict__ block_cones, const GrtCone* __restrict__ super_cones,
                                                              const uint32_t* __restrict__ starts, const uint32_t* __restrict__ counts,
                                                              uint32_t capacity, BinOut out) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t off = 0u, end = 0u;
    if (p < bvh.N && flag[0] != 0u) {
        off = min(starts[p], capacity);
        end = min(off + counts[p], capacity);
    }
    const bool have = end > off;
    float4 a = make_float4(1.f, 0.f, 0.f, 0.f), b = make_float4(1.f, 0.f, 0.f, 0.f), e = make_float4(1.f, 0.f, 0.f, 0.f);
    const f3 o = world_origin(P, mk3(ray_o[0], ray_o[1], ray_o[2]));
    const float dmin = __uint_as_float(dir_len_enc[0]), dmax = __uint_as_float(dir_len_enc[1]);
    if (have) {
        const float4* rec = reinterpret_cast<const float4*>(bvh.inst) + 3 * (size_t)p;
        a = rec[0]; b = rec[1]; e = rec[2];
    }
    const uint32_t np = have ? out.pair_n[p] : 0u;
    const uint32_t gx = blocks_x(P.W), gy = blocks_y(P.H);
    const GrtGrid G = grt_block_grid(block_cones, gx * gy, gx);
    if (__float_as_uint(G.hdr[9]) != 0u) {   // lists over the frame's tangent plane: a lane writes the cells its counting pass found
        if (have && np != kGridWaveTested) {
            const uint4 pr = out.pairs[(size_t)p * kBinCachedPairs];
            const uint32_t bx0 = pr.x & 0xFFFu, by0 = (pr.x >> 12) & 0xFFFu, wdt = (pr.x >> 24) + 1u;
            const float iw = 1.f / (float)wdt;
            unsigned long long hm = (unsigned long long)pr.y | ((unsigned long long)pr.z << 32);
            while (hm && off < end) {
                const int j = __ffsll((long long)hm) - 1;
                hm &= hm - 1ull;
                const uint32_t ry = (uint32_t)(((float)j + 0.5f) * iw), rx = (uint32_t)j - ry * wdt;
                out.block_keys[off] = (by0 + ry) * gx + bx0 + rx;
                out.vals[off] = p;
                ++off;
            }
        }
        const bool again = have && np == kGridWaveTested;
        if (__any(again)) {
            const BinParticle q = bin_particle(a, b, e, o, dmin, dmax, P.prim, p, P.box8);
            GridParticle g = grid_particle(G, gx, gy, again, q, P.bin_lane_area);
            if (!again) g.kind = 0;
            uint32_t n = 0u;
            grid_wave_cells<true>(P, G, block_cones, super_cones, lane, g, q, p, n, off, end, out);
        }
        for (; off < end; ++off) { out.block_keys[off] = 0xFFFFFFFFu; out.vals[off] = 0xFFFFFFFFu; }   // (not expected)
        return;
    }
    const bool cached = have && np <= (uint32_t)kBinCachedPairs;
    emit_cached_pairs(P, lane, cached, p, np, off, end, out);
    // (what the cache did not hold: tested again; pads whatever the masks left unwritten, which is not expected)
    const BinParticle q = bin_particle(a, b, e, o, dmin, dmax, P.prim, p, P.box8);
    const bool again = have && !cached;
    if (__any(again)) bin_pairs<true>(P, block_cones, super_cones, lane, again, q, p, off, again ? end : off, out);
    if (cached) {
        for (; off < end; ++off) { out.block_keys[off] = 0xFFFFFFFFu; out.vals[off] = 0xFFFFFFFFu; }
    }
}
__global__ __launch_bounds__(256) void grt_list_ranges_kernel(uint32_t n_max, const uint32_t* __restrict__ n_dev, uint32_t num_blocks,
                                                              const uint32_t* __restrict__ sorted_keys, uint32_t* __restrict__ ranges) {
    // four consecutive keys per thread (one 16-byte load + the key in front of them); n_dev: the entry count, read on the device
    // (a speculative launch against the buffers' capacity n_max)
    const uint32_t n = n_dev ? min(*n_dev, n_max) : n_max;
    const uint32_t e0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
    if (e0 >= n) return;
    uint32_t key[4];
    if (e0 + 4u <= n) {
        const uint4 v = *reinterpret_cast<const uint4*>(sorted_keys + e0);
        key[0] = v.x; key[1] = v.y; key[2] = v.z; key[3] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) key[j] = (e0 + j < n) ? sorted_keys[e0 + j] : 0xFFFFFFFFu;
    }
    uint32_t prev = e0 ? sorted_keys[e0 - 1] : 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t e = e0 + j;
        if (e >= n) break;
        const uint32_t k = key[j];
        if (k != prev || e == 0) {
            if (e != 0 && prev < num_blocks) ranges[2 * (size_t)prev + 1] = e;
            if (k < num_blocks) ranges[2 * (size_t)k] = e;
        }
        if (e == n - 1 && k < num_blocks) ranges[2 * (size_t)k + 1] = n;
        prev = k;
    }
}
void grt_launch_list_cones(hipStream_t s, const GrtTraceParams& P, const float* ray_o, const float* ray_d, uint32_t* uniform_origin,
                           uint32_t* dir_len_enc, GrtCone* block_cones, GrtCone* super_cones) {
    const GrtGrid G = grt_block_grid(block_cones, grt_num_blocks(P.W, P.H), blocks_x(P.W));
    const bool no_grid = getenv("GRUT_GRT_NO_GRID") != nullptr;   // (development switch, read per frame: the super tile scan for every frame)
    const bool packable = blocks_x(P.W) <= 4096u && blocks_y(P.H) <= 4096u;   // grid_pack keeps 12 bits per coordinate (frames up to 32768 pixels a side)
    hipLaunchKernelGGL(grt_list_init_kernel, dim3(1), dim3(1), 0, s, uniform_origin, dir_len_enc, G.hdr, (no_grid || P.sphere_lists || !packable) ? 0u : 1u);
    hipLaunchKernelGGL(grt_block_cone_kernel, dim3(grt_num_blocks(P.W, P.H)), dim3(64), 0, s, P, ray_o, ray_d, uniform_origin, block_cones);
    hipLaunchKernelGGL(grt_grid_tables_kernel, dim3(blocks_x(P.W) + blocks_y(P.H)), dim3(64), 0, s, P, block_cones);
    hipLaunchKernelGGL(grt_super_cone_kernel, dim3(grt_num_super(P.W, P.H)), dim3(64), 0, s, P, block_cones, super_cones, dir_len_enc);
}
void grt_launch_list_count(hipStream_t s, const GrtTraceParams& P, const GrtBvh& bvh, const float* ray_o, const uint32_t* uniform_origin,
                           const uint32_t* dir_len_enc, const GrtCone* block_cones, const GrtCone* super_cones, float* inst_rel, uint32_t* key_bits,
                           uint32_t* counts, uint32_t* particle_idx, void* pair_cache) {
    hipLaunchKernelGGL(grt_list_count_kernel, dim3(div_up(bvh.N, 256)), dim3(256), 0, s, P, bvh, ray_o, uniform_origin, dir_len_enc, block_cones,
                       super_cones, inst_rel, key_bits, counts, particle_idx, reinterpret_cast<uint4*>(pair_cache),
                       reinterpret_cast<uint32_t*>(reinterpret_cast<uint4*>(pair_cache) + (size_t)bvh.N * kBinCachedPairs));
}
void grt_launch_list_expand(hipStream_t s, const GrtTraceParams& P, const GrtBvh& bvh, const float* ray_o, const uint32_t* uniform_origin,
                            const uint32_t* dir_len_enc, const GrtCone* block_cones, const GrtCone* super_cones, const uint32_t* rank_to_particle,
                            const uint32_t* offsets, const uint32_t* counts, uint32_t* starts, uint32_t capacity, uint32_t* block_keys, uint32_t* vals,
                            void* pair_cache) {
    const BinOut out = {block_keys, vals, reinterpret_cast<uint4*>(pair_cache),
                        reinterpret_cast<uint32_t*>(reinterpret_cast<uint4*>(pair_cache) + (size_t)bvh.N * kBinCachedPairs)};
    hipLaunchKernelGGL(grt_list_starts_kernel, dim3(div_up(bvh.N, 256)), dim3(256), 0, s, bvh.N, rank_to_particle, offsets, starts);
    hipLaunchKernelGGL(grt_list_expand_kernel, dim3(div_up(bvh.N, 256)), dim3(256), 0, s, P, bvh, ray_o, uniform_origin, dir_len_enc, block_cones,
                       super_cones, starts, counts, capacity, out);
}
void grt_launch_list_ranges(hipStream_t s, uint32_t n, const uint32_t* n_dev, uint32_t num_blocks, const uint32_t* sorted_keys, uint32_t* ranges) {
    if (n == 0) return;
    hipLaunchKernelGGL(grt_list_ranges_kernel, dim3(div_up(div_up(n, 4u), 256)), dim3(256), 0, s, n, n_dev, num_blocks, sorted_keys, ranges);
}

void grt_launch_nht_fwd(hipStream_t s, const GrtTraceParams& P, const float* density12, const float* features, const float* ray_o, const float* ray_d,
                        float* out_feat, const GrtHitLog& log) {
    const dim3 grid(pixel_block_grid(P.W, P.H));
    const bool fast = P.nht_k == 48 && P.nht_ipd == 12 && P.nht_support == 1 && P.nht_act == 2 && P.nht_nf == 1 && !getenv("GRUT_NHT_GENERIC");
    if (fast) {
        GRT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((grt_nht_fwd_kernel<D_, true>), grid, dim3(64), 0, s, P, reinterpret_cast<const float4*>(density12), features,
                                                         ray_o, ray_d, out_feat, log));
    } else {
        GRT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((grt_nht_fwd_kernel<D_>), grid, dim3(64), 0, s, P, reinterpret_cast<const float4*>(density12), features,
                                                         ray_o, ray_d, out_feat, log));
    }
}

void grt_launch_trace_fwd(hipStream_t s, const GrtTraceParams& P, const GrtBvh& bvh, const float* density12, const float* sph,
                          const float* ray_o, const float* ray_d, float* out_rad, float* out_dns, float* out_hit2, float* out_nrm,
                          float* out_cnt, int32_t* visibility, uint32_t* dbg_ids, uint32_t* dbg_count, unsigned long long* counters,
                          const GrtHitLog& log, const GrtLists& lists) {
    const bool uni = lists.ranges != nullptr;   // the host knows by now whether the frame has one ray origin (it sized the lists)
    const dim3 grid(pixel_block_grid(P.W, P.H));
    const bool plain = P.prim == GRUT_PRIM_INSTANCES && !P.nht && !P.sph_half && !P.out_half && !P.normals;   // -> GEN = false
#define GRT_FWD_LAUNCH_G(COUNT_, UNI_, LOG_, GEN_)                                                                                                \
    GRT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((grt_trace_fwd_kernel<D_, COUNT_, UNI_, LOG_, GEN_>), grid, dim3(64), 0, s, P, bvh,         \
                                                     reinterpret_cast<const float4*>(density12), sph, ray_o, ray_d, out_rad, out_dns, out_hit2, \
                                                     out_nrm, out_cnt, visibility, dbg_ids, dbg_count, counters, log, lists))
#define GRT_FWD_LAUNCH(COUNT_, UNI_, LOG_)                                                                 \
    if (plain) { GRT_FWD_LAUNCH_G(COUNT_, UNI_, LOG_, false); } else { GRT_FWD_LAUNCH_G(COUNT_, UNI_, LOG_, true); }
    // (the instrumented build exists for the default configuration only: the counters are a development aid)
    if (counters && log.pool) {
        if (uni) { GRT_FWD_LAUNCH(true, true, true); } else { GRT_FWD_LAUNCH(true, false, true); }
    } else if (counters) {
        if (uni) { GRT_FWD_LAUNCH(true, true, false); } else { GRT_FWD_LAUNCH(true, false, false); }
    } else if (log.pool) {
        if (uni) { GRT_FWD_LAUNCH(false, true, true); } else { GRT_FWD_LAUNCH(false, false, true); }
    } else {
        if (uni) { GRT_FWD_LAUNCH(false, true, false); } else { GRT_FWD_LAUNCH(false, false, false); }
    }
#undef GRT_FWD_LAUNCH
}
void grt_launch_trace_bwd(hipStream_t s, hipStream_t s_rederive, const GrtTraceParams& P, const GrtBvh& bvh, const float* density12, const float* sph,
                          const float* ray_o, const float* ray_d, const float* rad, const float* dns, const float* hit2, const float* g_rad,
                          const float* g_dns, const float* g_hit, float* g_density12, float* g_sph, const GrtHitLog& log, const GrtLists& lists) {
    const dim3 grid(pixel_block_grid(P.W, P.H));
    // the re-derivation first: with a log it serves a handful of rays (every ray if the log overflowed) and runs, on its own stream,
    // next to the replay that follows
#define GRT_BWD_LAUNCH(UNI_)                                                                                                                     \
    GRT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((grt_trace_bwd_kernel<D_, UNI_>), grid, dim3(64), 0, s_rederive, P, bvh,                   \
                                                     reinterpret_cast<const float4*>(density12), sph, ray_o, ray_d, rad, dns, hit2, g_rad, g_dns, \
                                                     g_hit, g_density12, g_sph, log.pool ? log.state : nullptr, log.pool ? log.ray_flags : nullptr, lists))
#define GRT_BWD_LAUNCH_NHT(UNI_)                                                                                                                 \
    GRT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((grt_trace_bwd_kernel<D_, UNI_, true>), grid, dim3(64), 0, s_rederive, P, bvh,             \
                                                     reinterpret_cast<const float4*>(density12), sph, ray_o, ray_d, rad, dns, hit2, g_rad, g_dns, \
                                                     g_hit, g_density12, g_sph, log.pool ? log.state : nullptr, log.pool ? log.ray_flags : nullptr, lists))
    if (P.nht) {   // neural harmonic features: the Slang backward pipeline (sph / g_sph = feature buffer and its gradient, rad / g_rad = [H,W,ray_dim])
        if (lists.ranges) { GRT_BWD_LAUNCH_NHT(true); } else { GRT_BWD_LAUNCH_NHT(false); }
        const bool fast = P.nht_k == 48 && P.nht_ipd == 12 && P.nht_support == 1 && P.nht_act == 2 && P.nht_nf == 1 && !getenv("GRUT_NHT_GENERIC");
        if (log.pool && fast) {
            GRT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((grt_replay_nht_bwd_kernel<D_, true>), grid, dim3(64), 0, s, P,
                                                             reinterpret_cast<const float4*>(density12), sph, ray_o, ray_d, rad, dns, hit2, g_rad,
                                                             g_dns, g_hit, g_density12, g_sph, log, bvh.inst, bvh.scene));
        } else if (log.pool) {
            GRT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((grt_replay_nht_bwd_kernel<D_>), grid, dim3(64), 0, s, P,
                                                             reinterpret_cast<const float4*>(density12), sph, ray_o, ray_d, rad, dns, hit2, g_rad,
                                                             g_dns, g_hit, g_density12, g_sph, log, bvh.inst, bvh.scene));
        }
        return;
    }
#undef GRT_BWD_LAUNCH_NHT
    if (lists.ranges) { GRT_BWD_LAUNCH(true); } else { GRT_BWD_LAUNCH(false); }
    if (log.pool && P.prim == GRUT_PRIM_INSTANCES) {
        GRT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((grt_replay_bwd_kernel<D_, false>), grid, dim3(64), 0, s, P,
                                                         reinterpret_cast<const float4*>(density12), sph, ray_o, ray_d, rad, dns, hit2, g_rad,
                                                         g_dns, g_hit, g_density12, g_sph, log, bvh.inst, bvh.scene));
    } else if (log.pool) {
        GRT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((grt_replay_bwd_kernel<D_, true>), grid, dim3(64), 0, s, P,
                                                         reinterpret_cast<const float4*>(density12), sph, ray_o, ray_d, rad, dns, hit2, g_rad,
                                                         g_dns, g_hit, g_density12, g_sph, log, bvh.inst, bvh.scene));
    }
#undef GRT_BWD_LAUNCH
}

void grt_launch_mesh_aabb(hipStream_t s, uint32_t F, const float* vertices, const int32_t* triangles, float* aabb, float* slack,
                          uint32_t* scene_enc) {
    hipLaunchKernelGGL(grt_scene_init_kernel, dim3(1), dim3(64), 0, s, scene_enc);
    hipLaunchKernelGGL(grt_mesh_aabb_kernel, dim3(div_up(F, 256)), dim3(256), 0, s, F, vertices, triangles, aabb, slack, scene_enc);
}
void grt_launch_hybrid(hipStream_t s, const GrtTraceParams& P, const GrtBvh& bvh, const GrtMeshView& mesh, const GrtHybridParams& hp,
                       const float* density12, const float* sph, const float* ray_o, const float* ray_d, const float* ray_max_t, float* out_rgb,
                       float* out_alpha, float* out_last_ray, uint32_t* out_bounces, const GrtLists& lists) {
    const dim3 grid(pixel_block_grid(P.W, P.H));
    if (P.prim == GRUT_PRIM_INSTANCES && !P.sph_half) {
        GRT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((grt_hybrid_kernel<D_, false>), grid, dim3(64), 0, s, P, bvh, mesh, hp,
                                                         reinterpret_cast<const float4*>(density12), sph, ray_o, ray_d, ray_max_t, out_rgb, out_alpha,
                                                         out_last_ray, out_bounces, lists));
    } else {
        GRT_DISPATCH_DEGREE(P.degree, hipLaunchKernelGGL((grt_hybrid_kernel<D_, true>), grid, dim3(64), 0, s, P, bvh, mesh, hp,
                                                         reinterpret_cast<const float4*>(density12), sph, ray_o, ray_d, ray_max_t, out_rgb, out_alpha,
                                                         out_last_ray, out_bounces, lists));
    }
}

}  // namespace grut

// optim.hip — visibility-masked Adam update of the Gaussian parameter tensors, all parameter groups in ONE launch.
//
// Replaces threedgrut/optimizers/optimizers.cu:49-77 (selective_adam_update_kernel) and the per-group launches of
// threedgrut/optimizers/__init__.py:85-124: same arithmetic (no bias correction, rows with visibility == 0 keep their
// parameter AND their moments), but
//   * one launch walks every group (positions, density, rotation, scale, albedo, specular) instead of six,
//   * each lane moves 16 bytes per stream (float4 over the flat [N*M] array; a float4 may straddle two rows, each
//     element looks up its own row's flag),
//   * a float4 whose rows are all invisible costs one flag read and nothing else.
// HBM-bound: 28 B per visible element (read p, g, m, v; write p, m, v), DESIGN.md §7.
#include "common.hpp"

namespace grut {

constexpr int kAdamMaxGroups = 8;
constexpr int kAdamThreads = 256;

struct AdamLaunch {
    GrutAdamGroup g[kAdamMaxGroups];
    uint32_t block_end[kAdamMaxGroups];  // exclusive prefix of blocks per group
    int num_groups;
    uint32_t rows;
    const void* visibility;
    uint32_t vis_mask;  // 0: all rows visible; otherwise AND-mask applied to the flag word / byte
    int vis_bytes;
};

__device__ __forceinline__ bool row_visible(const AdamLaunch& L, uint32_t row) {
    if (L.vis_mask == 0u) return true;
    if (L.vis_bytes == 1) return (reinterpret_cast<const uint8_t*>(L.visibility)[row] & L.vis_mask) != 0;
    return (reinterpret_cast<const uint32_t*>(L.visibility)[row] & L.vis_mask) != 0u;
}

// optimizers.cu:62-75, operation order kept ((1-b2)*g)*g, step = -lr*m/(sqrt(v)+eps)
__device__ __forceinline__ void adam_element(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps) {
    m = b1 * m + (1.0f - b1) * g;
    v = b2 * v + (1.0f - b2) * g * g;
    const float step = -lr * m / (sqrtf(v) + eps);
    p += step;
}

__global__ __launch_bounds__(kAdamThreads) void selective_adam_kernel(const AdamLaunch L) {
    int gi = 0;
    while (gi + 1 < L.num_groups && blockIdx.x >= L.block_end[gi]) ++gi;
    const GrutAdamGroup G = L.g[gi];
    const uint32_t first_block = gi == 0 ? 0u : L.block_end[gi - 1];
    const uint64_t total = (uint64_t)L.rows * G.row_width;
    const uint64_t base = ((uint64_t)(blockIdx.x - first_block) * kAdamThreads + threadIdx.x) * 4ull;
    if (base >= total) return;
    const uint32_t M = G.row_width;
    uint32_t row = (uint32_t)(base / M);
    uint32_t rem = (uint32_t)(base - (uint64_t)row * M);
    bool vis[4];
    bool any = false;
    const int cnt = (total - base) >= 4ull ? 4 : (int)(total - base);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        vis[k] = false;
        if (k < cnt) {
            vis[k] = (k > 0 && rem != 0u) ? vis[k - 1] : row_visible(L, row);   // same row as the previous element
            any |= vis[k];
            if (++rem == M) { rem = 0u; ++row; }
        }
    }
    if (!any) return;
    if (cnt == 4) {
        float4 p = *reinterpret_cast<const float4*>(G.param + base);
        const float4 g = *reinterpret_cast<const float4*>(G.grad + base);
        float4 m = *reinterpret_cast<const float4*>(G.exp_avg + base);
        float4 v = *reinterpret_cast<const float4*>(G.exp_avg_sq + base);
        if (vis[0]) adam_element(p.x, g.x, m.x, v.x, G.lr, G.beta1, G.beta2, G.eps);
        if (vis[1]) adam_element(p.y, g.y, m.y, v.y, G.lr, G.beta1, G.beta2, G.eps);
        if (vis[2]) adam_element(p.z, g.z, m.z, v.z, G.lr, G.beta1, G.beta2, G.eps);
        if (vis[3]) adam_element(p.w, g.w, m.w, v.w, G.lr, G.beta1, G.beta2, G.eps);
        *reinterpret_cast<float4*>(G.param + base) = p;
        *reinterpret_cast<float4*>(G.exp_avg + base) = m;
        *reinterpret_cast<float4*>(G.exp_avg_sq + base) = v;
    } else {
        for (int k = 0; k < cnt; ++k) {
            if (!vis[k]) continue;
            float p = G.param[base + k], m = G.exp_avg[base + k], v = G.exp_avg_sq[base + k];
            adam_element(p, G.grad[base + k], m, v, G.lr, G.beta1, G.beta2, G.eps);
            G.param[base + k] = p;
            G.exp_avg[base + k] = m;
            G.exp_avg_sq[base + k] = v;
        }
    }
}

// ---- parameter marshalling ---------------------------------------------------------------------
// The renderers consume ParticleDensity rows {position, density, quaternion, scale, pad} (48 B); the model keeps the four
// tensors apart and the reference's plugin concatenates them with torch.cat every call (threedgut_tracer/tracer.py:178,
// threedgrt_tracer/tracer.py:93-96).  One pass, 44 B in / 48 B out per particle; a lane writes one float4 of a row.
__global__ __launch_bounds__(256) void pack_particles_kernel(uint32_t n, const float* __restrict__ pos, const float* __restrict__ dns,
                                                             const float* __restrict__ rot, const float* __restrict__ scl,
                                                             float4* __restrict__ out) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;   // float4 index: row = q / 3, part = q % 3
    if (q >= 3u * n) return;
    const uint32_t i = q / 3u, part = q - 3u * i;
    float4 v;
    if (part == 0u) v = make_float4(pos[3 * (size_t)i], pos[3 * (size_t)i + 1], pos[3 * (size_t)i + 2], dns[i]);
    else if (part == 1u) v = reinterpret_cast<const float4*>(rot)[i];
    else v = make_float4(scl[3 * (size_t)i], scl[3 * (size_t)i + 1], scl[3 * (size_t)i + 2], 0.f);
    out[q] = v;
}

// the inverse of pack_particles for gradients: [N,12] -> four contiguous tensors (autograd's AccumulateGrad then keeps
// them as they are; handing it strided views of the packed gradient costs one clone kernel per tensor)
__global__ __launch_bounds__(256) void unpack_grads_kernel(uint32_t n, const float4* __restrict__ g_packed, float* __restrict__ g_pos,
                                                           float* __restrict__ g_dns, float* __restrict__ g_rot, float* __restrict__ g_scl) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= 3u * n) return;
    const uint32_t i = q / 3u, part = q - 3u * i;
    const float4 g = g_packed[q];
    if (part == 0u) {
        g_pos[3 * (size_t)i] = g.x; g_pos[3 * (size_t)i + 1] = g.y; g_pos[3 * (size_t)i + 2] = g.z;
        g_dns[i] = g.w;
    } else if (part == 1u) {
        reinterpret_cast<float4*>(g_rot)[i] = g;
    } else {
        g_scl[3 * (size_t)i] = g.x; g_scl[3 * (size_t)i + 1] = g.y; g_scl[3 * (size_t)i + 2] = g.z;
    }
}

// Activations of the model fused with the packing (threedgrut/model/model.py:102-118, utils/misc.py:44-49):
// density = sigmoid(raw), scale = exp(raw), rotation = raw / max(|raw|, 1e-12) (torch.nn.functional.normalize).
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__global__ __launch_bounds__(256) void activate_pack_kernel(uint32_t n, const float* __restrict__ pos, const float* __restrict__ raw_dns,
                                                            const float* __restrict__ raw_rot, const float* __restrict__ raw_scl,
                                                            float4* __restrict__ out) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= 3u * n) return;
    const uint32_t i = q / 3u, part = q - 3u * i;
    float4 v;
    if (part == 0u) {
        v = make_float4(pos[3 * (size_t)i], pos[3 * (size_t)i + 1], pos[3 * (size_t)i + 2], sigmoidf_(raw_dns[i]));
    } else if (part == 1u) {
        const float4 r = reinterpret_cast<const float4*>(raw_rot)[i];
        const float inv = 1.f / fmaxf(sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w), 1e-12f);
        v = make_float4(r.x * inv, r.y * inv, r.z * inv, r.w * inv);
    } else {
        v = make_float4(expf(raw_scl[3 * (size_t)i]), expf(raw_scl[3 * (size_t)i + 1]), expf(raw_scl[3 * (size_t)i + 2]), 0.f);
    }
    out[q] = v;
}
// chain rule back to the raw parameters, split into the model's four tensors (contiguous: the optimizer's AccumulateGrad
// takes them without a copy)
__global__ __launch_bounds__(256) void activate_pack_bwd_kernel(uint32_t n, const float* __restrict__ raw_dns, const float* __restrict__ raw_rot,
                                                                const float* __restrict__ raw_scl, const float4* __restrict__ g_packed,
                                                                float* __restrict__ g_pos, float* __restrict__ g_dns,
                                                                float* __restrict__ g_rot, float* __restrict__ g_scl) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= 3u * n) return;
    const uint32_t i = q / 3u, part = q - 3u * i;
    const float4 g = g_packed[q];
    if (part == 0u) {
        g_pos[3 * (size_t)i] = g.x; g_pos[3 * (size_t)i + 1] = g.y; g_pos[3 * (size_t)i + 2] = g.z;
        const float s = sigmoidf_(raw_dns[i]);
        g_dns[i] = g.w * s * (1.f - s);
    } else if (part == 1u) {
        const float4 r = reinterpret_cast<const float4*>(raw_rot)[i];
        const float len = sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w);
        float4 o;
        if (len > 1e-12f) {   // d(r/|r|) = (g - n (n.g)) / |r|
            const float inv = 1.f / len;
            const float nx = r.x * inv, ny = r.y * inv, nz = r.z * inv, nw = r.w * inv;
            const float d = nx * g.x + ny * g.y + nz * g.z + nw * g.w;
            o = make_float4((g.x - nx * d) * inv, (g.y - ny * d) * inv, (g.z - nz * d) * inv, (g.w - nw * d) * inv);
        } else {              // clamped denominator: the map is linear there
            o = make_float4(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f, g.w * 1e12f);
        }
        reinterpret_cast<float4*>(g_rot)[i] = o;
    } else {
        g_scl[3 * (size_t)i] = g.x * expf(raw_scl[3 * (size_t)i]);
        g_scl[3 * (size_t)i + 1] = g.y * expf(raw_scl[3 * (size_t)i + 1]);
        g_scl[3 * (size_t)i + 2] = g.z * expf(raw_scl[3 * (size_t)i + 2]);
    }
}

}  // namespace grut

extern "C" int grut_pack_particles(void* stream, uint32_t num_particles, const float* positions, const float* density,
                                   const float* rotation, const float* scale, float* particle_density) {
    using namespace grut;
    if (num_particles == 0) return GRUT_OK;
    GRUT_REQUIRE(positions && density && rotation && scale && particle_density, "grut_pack_particles: null tensor");
    GRUT_REQUIRE(((uintptr_t)rotation | (uintptr_t)particle_density) % 16 == 0, "grut_pack_particles: rotation / output must be 16-byte aligned");
    GRUT_REQUIRE(num_particles <= 0x55555555u, "grut_pack_particles: too many particles");
    const uint32_t quads = 3u * num_particles;
    hipLaunchKernelGGL(pack_particles_kernel, dim3((quads + 255u) / 256u), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), num_particles,
                       positions, density, rotation, scale, reinterpret_cast<float4*>(particle_density));
    GRUT_HIP(hipGetLastError());
    return GRUT_OK;
}

extern "C" int grut_unpack_particle_grads(void* stream, uint32_t num_particles, const float* grad_particle_density, float* grad_positions,
                                         float* grad_density, float* grad_rotation, float* grad_scale) {
    using namespace grut;
    if (num_particles == 0) return GRUT_OK;
    GRUT_REQUIRE(grad_particle_density && grad_positions && grad_density && grad_rotation && grad_scale, "grut_unpack_particle_grads: null tensor");
    GRUT_REQUIRE(((uintptr_t)grad_particle_density | (uintptr_t)grad_rotation) % 16 == 0,
                 "grut_unpack_particle_grads: packed / rotation gradients must be 16-byte aligned");
    GRUT_REQUIRE(num_particles <= 0x55555555u, "grut_unpack_particle_grads: too many particles");
    const uint32_t quads = 3u * num_particles;
    hipLaunchKernelGGL(unpack_grads_kernel, dim3((quads + 255u) / 256u), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), num_particles,
                       reinterpret_cast<const float4*>(grad_particle_density), grad_positions, grad_density, grad_rotation, grad_scale);
    GRUT_HIP(hipGetLastError());
    return GRUT_OK;
}

extern "C" int grut_activate_pack(void* stream, uint32_t num_particles, const float* positions, const float* raw_density,
                                  const float* raw_rotation, const float* raw_scale, float* particle_density) {
    using namespace grut;
    if (num_particles == 0) return GRUT_OK;
    GRUT_REQUIRE(positions && raw_density && raw_rotation && raw_scale && particle_density, "grut_activate_pack: null tensor");
    GRUT_REQUIRE(((uintptr_t)raw_rotation | (uintptr_t)particle_density) % 16 == 0, "grut_activate_pack: rotation / output must be 16-byte aligned");
    GRUT_REQUIRE(num_particles <= 0x55555555u, "grut_activate_pack: too many particles");
    const uint32_t quads = 3u * num_particles;
    hipLaunchKernelGGL(activate_pack_kernel, dim3((quads + 255u) / 256u), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), num_particles,
                       positions, raw_density, raw_rotation, raw_scale, reinterpret_cast<float4*>(particle_density));
    GRUT_HIP(hipGetLastError());
    return GRUT_OK;
}

extern "C" int grut_activate_pack_backward(void* stream, uint32_t num_particles, const float* raw_density, const float* raw_rotation,
                                           const float* raw_scale, const float* grad_particle_density, float* grad_positions,
                                           float* grad_raw_density, float* grad_raw_rotation, float* grad_raw_scale) {
    using namespace grut;
    if (num_particles == 0) return GRUT_OK;
    GRUT_REQUIRE(raw_density && raw_rotation && raw_scale && grad_particle_density && grad_positions && grad_raw_density && grad_raw_rotation &&
                     grad_raw_scale, "grut_activate_pack_backward: null tensor");
    GRUT_REQUIRE(((uintptr_t)raw_rotation | (uintptr_t)grad_particle_density | (uintptr_t)grad_raw_rotation) % 16 == 0,
                 "grut_activate_pack_backward: rotation / packed gradient tensors must be 16-byte aligned");
    GRUT_REQUIRE(num_particles <= 0x55555555u, "grut_activate_pack_backward: too many particles");
    const uint32_t quads = 3u * num_particles;
    hipLaunchKernelGGL(activate_pack_bwd_kernel, dim3((quads + 255u) / 256u), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), num_particles,
                       raw_density, raw_rotation, raw_scale, reinterpret_cast<const float4*>(grad_particle_density), grad_positions,
                       grad_raw_density, grad_raw_rotation, grad_raw_scale);
    GRUT_HIP(hipGetLastError());
    return GRUT_OK;
}

extern "C" int grut_selective_adam_update(void* stream, const GrutAdamGroup* groups, int num_groups, uint32_t num_rows,
                                          const void* visibility, int visibility_kind) {
    using namespace grut;
    GRUT_REQUIRE(num_groups >= 0 && (num_groups == 0 || groups), "grut_selective_adam_update: groups is NULL");
    GRUT_REQUIRE(visibility_kind >= GRUT_VIS_NONE && visibility_kind <= GRUT_VIS_FLOAT_BITS, "grut_selective_adam_update: visibility_kind %d", visibility_kind);
    GRUT_REQUIRE(visibility_kind == GRUT_VIS_NONE || visibility || num_rows == 0, "grut_selective_adam_update: visibility is NULL");
    if (num_rows == 0) return GRUT_OK;  // optimizers.cu:87-89
    for (int first = 0; first < num_groups;) {  // `first` resumes where the previous batch stopped consuming (empty groups do not count)
        AdamLaunch L{};
        L.rows = num_rows;
        L.visibility = visibility;
        L.vis_bytes = visibility_kind == GRUT_VIS_BOOL_U8 ? 1 : 4;
        L.vis_mask = visibility_kind == GRUT_VIS_NONE ? 0u : visibility_kind == GRUT_VIS_BOOL_U8 ? 0xFFu
                   : visibility_kind == GRUT_VIS_FLOAT_BITS ? 0x7FFFFFFFu : 0xFFFFFFFFu;
        uint64_t blocks = 0;
        int i = first;
        for (; i < num_groups && L.num_groups < kAdamMaxGroups; ++i) {
            const GrutAdamGroup& g = groups[i];
            if (g.row_width == 0) continue;
            GRUT_REQUIRE(g.param && g.grad && g.exp_avg && g.exp_avg_sq, "grut_selective_adam_update: group %d has a NULL tensor", i);
            GRUT_REQUIRE(((uintptr_t)g.param | (uintptr_t)g.grad | (uintptr_t)g.exp_avg | (uintptr_t)g.exp_avg_sq) % 16 == 0,
                         "grut_selective_adam_update: group %d tensors must be 16-byte aligned", i);
            const uint64_t quads = ((uint64_t)num_rows * g.row_width + 3) / 4;
            blocks += (quads + kAdamThreads - 1) / kAdamThreads;
            GRUT_REQUIRE(blocks < 0x7FFFFFFFull, "grut_selective_adam_update: too many elements");
            L.g[L.num_groups] = g;
            L.block_end[L.num_groups] = (uint32_t)blocks;
            ++L.num_groups;
        }
        first = i;
        if (L.num_groups == 0) continue;
        hipLaunchKernelGGL(selective_adam_kernel, dim3((uint32_t)blocks), dim3(kAdamThreads), 0, reinterpret_cast<hipStream_t>(stream), L);
        GRUT_HIP(hipGetLastError());
    }
    return GRUT_OK;
}

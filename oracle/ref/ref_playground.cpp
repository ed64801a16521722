// ref_playground.cpp — runs the reference's OWN hybrid mesh + Gaussian path tracer on the host: __raygen__rg (the path loop),
// __closesthit__ch (normals, material dispatch: handleMirror / handleGlass / handleDiffuse / handlePBR), __miss__ms, __intersection__is
// and __anyhit__ah of threedgrut_playground/src/kernels/cuda/playgroundKernel.cu, included as they lie, with everything under them:
// playground/kernels/cuda/trace.cuh (traceMesh, traceGaussians, getBackgroundColor, the payload), materials.cuh (Cook-Torrance
// sampling, GGX importance sampling, textures, tangent frames, alpha test), rng.cuh, mathUtils.cuh and
// 3dgrt/kernels/cuda/3dgrtTracer.cuh (traceVolumetricGS: the k = 16 rounds; the any-hit insertion chain).
//
// What is NOT the reference's and is restated around it:
//   * OptiX itself (an un-vendored dependency): `optixTrace` on the triangle handle returns the closest triangle in (tmin, tmax) by
//     Moeller-Trumbore over every triangle — separately rounded fp32 operations in source order, ties to the lower face index (the
//     hardware's watertight test is not specified to the bit) — and runs the closest-hit or the miss program; on the particle handle it
//     offers the ray every proxy instance whose unit box overlaps the ray's interval (as ref_grt_emul.inl does for the trainer's
//     programs).  Nested traces (handleDiffuse traces Gaussians from inside the closest-hit program) save and restore the hit state.
//   * the Slang-generated per-hit header (shim/3dgrt_slang/..., restated from the .slang sources and cross-checked below against the
//     reference's hand-written CUDA twin) and CUDA texture objects (shim/playground_shim/playground/cutexture.h).
// TEST INFRASTRUCTURE ONLY: pins oracle/grt_oracle.c's hybrid tracer (tests/golden/playground.npz).
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>
#include "shim/cuda_shim.h"
#define __global__
#define __constant__
#define SHIM_OPTIX_DEVICE_API
#define SHIM_OPTIX_PLAYGROUND
#define SPH_MAX_NUM_COEFFS 16
#define PARTICLE_RADIANCE_NUM_COEFFS 16
#define GAUSSIAN_PARTICLE_KERNEL_DEGREE PARTICLE_KERNEL_DEGREE
#define GAUSSIAN_PARTICLE_MIN_KERNEL_DENSITY 0.0113f
#define GAUSSIAN_PARTICLE_MIN_ALPHA (1.0f / 255.0f)
#define GAUSSIAN_PARTICLE_MAX_ALPHA 0.99f
#define PARTICLE_FEATURE_DIM 48
#define RAY_FEATURE_DIM 3
#define FEATURE_TRANSFORM_TYPE 0
#define PARTICLE_PRIMITIVE_TYPE MOGPrimitiveTypes::MOGTracingInstances
#define PARTICLE_PRIMITIVE_CLAMPED 1
#define ENABLE_HIT_COUNTS
using std::isfinite;
#include "shim/optix.h"
thread_local ShimOptix g_optix;

#include "../_ref/playground_kernel.inc"

// ---- the emulated traversal ---------------------------------------------------------------------------------------------------
namespace {
template <class A, class T, int N>
void fill_acc(A& acc, T* data, const int32_t (&sizes)[N], const int32_t (&strides)[N]) {
    struct Raw { T* data; int32_t sizes[N]; int32_t strides[N]; } raw;
    raw.data = data;
    for (int i = 0; i < N; ++i) { raw.sizes[i] = sizes[i]; raw.strides[i] = strides[i]; }
    static_assert(sizeof(Raw) == sizeof(A), "PackedTensorAccessor32 layout");
    std::memcpy(&acc, &raw, sizeof(raw));
}
struct Scene {
    uint32_t n = 0;
    std::vector<float> inv;           // [n][12] inverse instance maps (rows of the inverse linear part, then the translation)
    uint32_t num_faces = 0;
    const float* vertices = nullptr;  // [V,3]
    const int32_t* faces = nullptr;   // [F,3]
    float trace_tmax = 0.f;
} g_scene;
constexpr OptixTraversableHandle kTriHandle = 2, kParticleHandle = 1;

// Moeller-Trumbore, every operation rounded separately (the library is built with -ffp-contract=off), source order as in
// oracle/grt_oracle.c: tri_intersect and csrc/grt_kernels.hip: tri_intersect
bool tri_hit(uint32_t f, float3 o, float3 d, float tmin, float tmax, float& t, float& u, float& v, float3 p[3]) {
    const int32_t* tr = g_scene.faces + 3 * (size_t)f;
    for (int k = 0; k < 3; ++k) p[k] = make_float3(g_scene.vertices[3 * (size_t)tr[k]], g_scene.vertices[3 * (size_t)tr[k] + 1], g_scene.vertices[3 * (size_t)tr[k] + 2]);
    const float e1x = p[1].x - p[0].x, e1y = p[1].y - p[0].y, e1z = p[1].z - p[0].z, e2x = p[2].x - p[0].x, e2y = p[2].y - p[0].y, e2z = p[2].z - p[0].z;
    const float pvx = d.y * e2z - d.z * e2y, pvy = d.z * e2x - d.x * e2z, pvz = d.x * e2y - d.y * e2x;
    const float det = e1x * pvx + e1y * pvy + e1z * pvz;
    if (!(fabsf(det) > 1e-20f)) return false;
    const float inv = 1.f / det;
    const float tvx = o.x - p[0].x, tvy = o.y - p[0].y, tvz = o.z - p[0].z;
    u = (tvx * pvx + tvy * pvy + tvz * pvz) * inv;
    if (u < 0.f || u > 1.f) return false;
    const float qx = tvy * e1z - tvz * e1y, qy = tvz * e1x - tvx * e1z, qz = tvx * e1y - tvy * e1x;
    v = (d.x * qx + d.y * qy + d.z * qz) * inv;
    if (v < 0.f || u + v > 1.f) return false;
    t = (e2x * qx + e2y * qy + e2z * qz) * inv;
    return (t > tmin) && (t < tmax);
}
}  // namespace

bool optixReportIntersection(float t, unsigned) {
    if (!(t >= g_optix.tmin && t <= g_optix.tmax)) return false;
    const float far_end = g_optix.tmax;
    g_optix.tmax = t;          // the any-hit program sees the reported distance as the ray's tmax
    g_optix.ignore = false;
    __anyhit__ah();
    if (g_optix.ignore) { g_optix.tmax = far_end; return false; }
    return true;
}

// the mesh pass: closest hit over the triangle GAS, any-hit disabled (trace.cuh:175-195)
void optixTrace(OptixTraversableHandle handle, float3 o, float3 d, float tmin, float tmax, float, OptixVisibilityMask, unsigned, unsigned, unsigned, unsigned,
                uint32_t& p0, uint32_t& p1) {
    if (handle != kTriHandle) abort();
    const ShimOptix saved = g_optix;
    g_optix.payload[0] = &p0; g_optix.payload[1] = &p1;
    g_optix.worldOrigin = o; g_optix.worldDirection = d;
    g_optix.tmin = tmin; g_optix.tmax = tmax;
    float best = 3.0e38f, bu = 0.f, bv = 0.f;
    int bf = -1;
    float3 bp[3] = {};
    for (uint32_t f = 0; f < g_scene.num_faces; ++f) {
        float t, u, v;
        float3 p[3];
        if (tri_hit(f, o, d, tmin, tmax, t, u, v, p) && t < best) { best = t; bu = u; bv = v; bf = (int)f; bp[0] = p[0]; bp[1] = p[1]; bp[2] = p[2]; }
    }
    if (bf >= 0) {
        g_optix.primitive = (unsigned)bf; g_optix.tmax = best; g_optix.barycentrics = make_float2(bu, bv);
        g_optix.triangle[0] = bp[0]; g_optix.triangle[1] = bp[1]; g_optix.triangle[2] = bp[2];
        __closesthit__ch();
    } else {
        __miss__ms();
    }
    const uint3 li = g_optix.launchIndex, ld = g_optix.launchDim;
    g_optix = saved;
    g_optix.launchIndex = li; g_optix.launchDim = ld;
}

// the particle pass: closest-hit disabled, the any-hit chain keeps the 16 nearest (3dgrtTracer.cuh:84-135)
void optixTrace(OptixTraversableHandle handle, float3 o, float3 d, float tmin, float tmax, float, OptixVisibilityMask, unsigned, unsigned, unsigned,
                unsigned, uint32_t& p0, uint32_t& p1, uint32_t& p2, uint32_t& p3, uint32_t& p4, uint32_t& p5, uint32_t& p6, uint32_t& p7,
                uint32_t& p8, uint32_t& p9, uint32_t& p10, uint32_t& p11, uint32_t& p12, uint32_t& p13, uint32_t& p14, uint32_t& p15,
                uint32_t& p16, uint32_t& p17, uint32_t& p18, uint32_t& p19, uint32_t& p20, uint32_t& p21, uint32_t& p22, uint32_t& p23,
                uint32_t& p24, uint32_t& p25, uint32_t& p26, uint32_t& p27, uint32_t& p28, uint32_t& p29, uint32_t& p30, uint32_t& p31) {
    if (handle != kParticleHandle) abort();
    const ShimOptix saved = g_optix;   // (this trace may run inside a closest-hit program: handleDiffuse)
    uint32_t* ps[32] = {&p0, &p1, &p2, &p3, &p4, &p5, &p6, &p7, &p8, &p9, &p10, &p11, &p12, &p13, &p14, &p15,
                        &p16, &p17, &p18, &p19, &p20, &p21, &p22, &p23, &p24, &p25, &p26, &p27, &p28, &p29, &p30, &p31};
    for (int k = 0; k < 32; ++k) g_optix.payload[k] = ps[k];
    g_optix.worldOrigin = o; g_optix.worldDirection = d;
    g_optix.tmin = tmin; g_optix.tmax = tmax;
    g_optix.primitive = 0;
    const float trace_tmax = tmax;
    for (uint32_t i = 0; i < g_scene.n; ++i) {
        const float* m = &g_scene.inv[12 * (size_t)i];
        const float dx = o.x - m[9], dy = o.y - m[10], dz = o.z - m[11];
        const float3 oo = make_float3(m[0] * dx + m[1] * dy + m[2] * dz, m[3] * dx + m[4] * dy + m[5] * dz, m[6] * dx + m[7] * dy + m[8] * dz);
        const float3 od = make_float3(m[0] * d.x + m[1] * d.y + m[2] * d.z, m[3] * d.x + m[4] * d.y + m[5] * d.z, m[6] * d.x + m[7] * d.y + m[8] * d.z);
        const float ax0 = (-1.f - oo.x) / od.x, ax1 = (1.f - oo.x) / od.x, ay0 = (-1.f - oo.y) / od.y, ay1 = (1.f - oo.y) / od.y;
        const float az0 = (-1.f - oo.z) / od.z, az1 = (1.f - oo.z) / od.z;
        const float tn = fmaxf(fmaxf(fminf(ax0, ax1), fminf(ay0, ay1)), fminf(az0, az1));
        const float tf = fminf(fminf(fmaxf(ax0, ax1), fmaxf(ay0, ay1)), fmaxf(az0, az1));
        if (!(tn <= tf) || !(tf >= g_optix.tmin) || !(tn <= trace_tmax)) continue;   // (box test against the interval the trace started with)
        g_optix.instance = i; g_optix.objectOrigin = oo; g_optix.objectDirection = od;
        __intersection__is();
    }
    g_optix = saved;
}

extern "C" {

int ref_playground_degree(void) { return PARTICLE_KERNEL_DEGREE; }

// One material of the reference's table (PBRMaterial, pipelineParameters.h:26-52) with its textures as host arrays ([H,W,C] f32; a
// NULL pointer = no texture).
struct RefMaterial {
    const float *diffuse_tex, *emissive_tex, *metallic_roughness_tex, *normal_tex;   // C = 4, 4, 2, 4
    int32_t diffuse_hw[2], emissive_hw[2], metallic_roughness_hw[2], normal_hw[2];
    float diffuse_factor[4], emissive_factor[3], metallic_factor, roughness_factor, transmission_factor, ior, alpha_cutoff;
    uint32_t alpha_mode;
};

// transforms [n,12]: instance matrices of the reference's instance kernel (ref_grt_proxies); rays [H*W,3] in world space, OVERWRITTEN
// with the last traced segment like the reference's buffers (trace.cuh:158-173); ray_max_t [H*W]; mesh arrays as the reference's
// tensors (triangles [F,3] i32, v_normals [V,3], v_tangents [V,3], v_has_tangents [V] u8, prim_type [F], mat_uv [F,3,2], mat_id [F],
// refractive_index [F]); envmap [EH,EW,4] or NULL; out_rgb [H*W,3], out_alpha [H*W].
void ref_playground_trace(uint32_t n, const float* transforms, const float* density12, const float* sph48, int width, int height, float* ray_o,
                          float* ray_d, const float* ray_max_t, const float* scene_aabb6, float min_transmittance, unsigned sph_degree,
                          unsigned frame_number, uint32_t num_vertices, const float* vertices, uint32_t num_faces, const int32_t* triangles,
                          const float* v_normals, const float* v_tangents, const uint8_t* v_has_tangents, const int32_t* prim_type,
                          const float* mat_uv, const int32_t* mat_id, const float* refractive_index, uint32_t num_materials,
                          const RefMaterial* materials, const float* envmap, int envmap_h, int envmap_w, const float* envmap_offset2,
                          unsigned playground_opts, unsigned max_pbr_bounces, float* out_rgb, float* out_alpha) {
    // instances: inverse maps
    g_scene.n = n;
    g_scene.inv.resize(12 * (size_t)n);
    for (uint32_t i = 0; i < n; ++i) {
        const float* t = &transforms[12 * (size_t)i];
        const double a = t[0], b = t[1], c = t[2], d = t[4], e = t[5], f = t[6], g = t[8], h = t[9], k = t[10];
        const double det = a * (e * k - f * h) - b * (d * k - f * g) + c * (d * h - e * g);
        const double inv[9] = {(e * k - f * h) / det, (c * h - b * k) / det, (b * f - c * e) / det, (f * g - d * k) / det, (a * k - c * g) / det,
                               (c * d - a * f) / det, (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det};
        float* m = &g_scene.inv[12 * (size_t)i];
        for (int q = 0; q < 9; ++q) m[q] = (float)inv[q];
        m[9] = t[3]; m[10] = t[7]; m[11] = t[11];
    }
    g_scene.num_faces = num_faces; g_scene.vertices = vertices; g_scene.faces = triangles;
    // launch parameters
    const int32_t s3[4] = {1, height, width, 3}, t3[4] = {height * width * 3, width * 3, 3, 1};
    const int32_t s1[4] = {1, height, width, 1}, t1[4] = {height * width, width, 1, 1};
    const int32_t s2[4] = {1, height, width, 2}, t2[4] = {height * width * 2, width * 2, 2, 1};
    std::vector<float> hitdist((size_t)width * height * 2, 0.f), normals((size_t)width * height * 3, 0.f), hits((size_t)width * height, 0.f);
    std::vector<int32_t> visibility(n ? n : 1, 0), trace_state((size_t)width * height, 0);
    params.rayToWorld[0] = make_float4(1, 0, 0, 0); params.rayToWorld[1] = make_float4(0, 1, 0, 0); params.rayToWorld[2] = make_float4(0, 0, 1, 0);
    fill_acc(params.rayOrigin, ray_o, s3, t3);
    fill_acc(params.rayDirection, ray_d, s3, t3);
    params.particleDensity = reinterpret_cast<const ParticleDensity*>(density12);
    params.particleFeatures = sph48;
    params.particleExtendedData = nullptr;
    params.particleVisibility = visibility.data();
    fill_acc(params.rayFeatures, out_rgb, s3, t3);
    fill_acc(params.rayDensity, out_alpha, s1, t1);
    fill_acc(params.rayHitDistance, hitdist.data(), s2, t2);
    fill_acc(params.rayNormal, normals.data(), s3, t3);
    fill_acc(params.rayHitsCount, hits.data(), s1, t1);
    params.handle = kParticleHandle;
    params.aabb = OptixAabb{scene_aabb6[0], scene_aabb6[1], scene_aabb6[2], scene_aabb6[3], scene_aabb6[4], scene_aabb6[5]};
    params.minTransmittance = min_transmittance;
    params.hitMinGaussianResponse = GAUSSIAN_PARTICLE_MIN_KERNEL_DENSITY;
    params.alphaMinThreshold = GAUSSIAN_PARTICLE_MIN_ALPHA;
    params.sphDegree = sph_degree;
    params.frameBounds = uint2{(unsigned)width - 1, (unsigned)height - 1};
    params.frameNumber = frame_number;
    params.gPrimNumTri = 0;
    // playground members
    const int32_t sm[3] = {1, height, width}, tm[3] = {height * width, width, 1};
    fill_acc(params.rayMaxT, const_cast<float*>(ray_max_t), sm, tm);
    ShimTexture env = {envmap_h, envmap_w, 4, envmap};
    params.envmap = reinterpret_cast<cudaTextureObject_t>(envmap ? &env : nullptr);
    params.envmapOffset = make_float2(envmap_offset2 ? envmap_offset2[0] : 0.f, envmap_offset2 ? envmap_offset2[1] : 0.f);
    params.triHandle = kTriHandle;
    params.playgroundOpts = playground_opts;
    params.maxPBRBounces = max_pbr_bounces;
    fill_acc(params.trace_state, trace_state.data(), s1, t1);
    const int32_t sf3[2] = {(int32_t)num_faces, 3}, tf3[2] = {3, 1}, sf1[2] = {(int32_t)num_faces, 1}, tf1[2] = {1, 1};
    const int32_t sv3[2] = {(int32_t)num_vertices, 3}, tv3[2] = {3, 1}, sv1[2] = {(int32_t)num_vertices, 1}, tv1[2] = {1, 1};
    fill_acc(params.triangles, const_cast<int32_t*>(triangles), sf3, tf3);
    fill_acc(params.vNormals, const_cast<float*>(v_normals), sv3, tv3);
    fill_acc(params.vHasTangents, reinterpret_cast<bool*>(const_cast<uint8_t*>(v_has_tangents)), sv1, tv1);
    fill_acc(params.vTangents, const_cast<float*>(v_tangents), sv3, tv3);
    const int32_t su[3] = {(int32_t)num_faces, 3, 2}, tu[3] = {6, 2, 1};
    fill_acc(params.matUV, const_cast<float*>(mat_uv), su, tu);
    fill_acc(params.matID, const_cast<int32_t*>(mat_id), sf1, tf1);
    fill_acc(params.primType, const_cast<int32_t*>(prim_type), sf1, tf1);
    fill_acc(params.refractiveIndex, const_cast<float*>(refractive_index), sf1, tf1);
    std::vector<PBRMaterial> mats(num_materials ? num_materials : 1);
    std::vector<ShimTexture> texs(4 * (size_t)(num_materials ? num_materials : 1));
    for (uint32_t i = 0; i < num_materials; ++i) {   // HybridOptixTracer::syncMaterials (hybridTracer.cpp:231-312)
        const RefMaterial& r = materials[i];
        PBRMaterial& m = mats[i];
        std::memset(&m, 0, sizeof(m));
        ShimTexture* t = &texs[4 * (size_t)i];
        t[0] = {r.diffuse_hw[0], r.diffuse_hw[1], 4, r.diffuse_tex};
        t[1] = {r.emissive_hw[0], r.emissive_hw[1], 4, r.emissive_tex};
        t[2] = {r.metallic_roughness_hw[0], r.metallic_roughness_hw[1], 2, r.metallic_roughness_tex};
        t[3] = {r.normal_hw[0], r.normal_hw[1], 4, r.normal_tex};
        m.useDiffuseTexture = r.diffuse_tex != nullptr; m.diffuseTexture = reinterpret_cast<cudaTextureObject_t>(r.diffuse_tex ? &t[0] : nullptr);
        m.useEmissiveTexture = r.emissive_tex != nullptr; m.emissiveTexture = reinterpret_cast<cudaTextureObject_t>(r.emissive_tex ? &t[1] : nullptr);
        m.useMetallicRoughnessTexture = r.metallic_roughness_tex != nullptr;
        m.metallicRoughnessTexture = reinterpret_cast<cudaTextureObject_t>(r.metallic_roughness_tex ? &t[2] : nullptr);
        m.useNormalTexture = r.normal_tex != nullptr; m.normalTexture = reinterpret_cast<cudaTextureObject_t>(r.normal_tex ? &t[3] : nullptr);
        m.diffuseFactor = make_float4(r.diffuse_factor[0], r.diffuse_factor[1], r.diffuse_factor[2], r.diffuse_factor[3]);
        m.emissiveFactor = make_float3(r.emissive_factor[0], r.emissive_factor[1], r.emissive_factor[2]);
        m.metallicFactor = r.metallic_factor; m.roughnessFactor = r.roughness_factor; m.transmissionFactor = r.transmission_factor;
        m.ior = r.ior; m.alphaCutoff = r.alpha_cutoff; m.alphaMode = r.alpha_mode;
    }
    params.materials = mats.data();
    params.numMaterials = num_materials;
    g_optix.launchDim = uint3{(unsigned)width, (unsigned)height, 1u};
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            g_optix.launchIndex = uint3{(unsigned)x, (unsigned)y, 0u};
            __raygen__rg();
        }
}

// cross-check of the Slang stand-in against the reference's hand-written CUDA twin is done by the golden script through
// libref_hit_deg4.so (same inputs through ref_hit.cpp's processHit): see tests/golden/make_golden.py: playground
float ref_playground_standin_process_hit(const float* ray_o3, const float* ray_d3, const float* density12, float* transmittance, float* depth) {
    gaussianParticle_CommonParameters_0 cp = {{(gaussianParticle_RawParameters_0*)density12, nullptr, true}};
    return particleDensityProcessHitFwdFromBuffer(make_float3(ray_o3[0], ray_o3[1], ray_o3[2]), make_float3(ray_d3[0], ray_d3[1], ray_d3[2]), 0u, cp,
                                                  transmittance, depth, false, nullptr);
}
void ref_playground_standin_integrate(const float* ray_d3, float weight, const float* sph48, unsigned sph_degree, float* integrated3) {
    particleFeaturesIntegrateFwdGeneric(make_float3(ray_d3[0], ray_d3[1], ray_d3[2]), weight, 0u, sph48, sph_degree, integrated3);
}

}  // extern "C"

// ref_gut_hit.cpp — C entry points over the REFERENCE's 3DGUT per-hit math (namespace threedgut), compiled on the host
// from threedgut_tracer/include/3dgut/kernels/cuda/models/gaussianParticles.cuh where it lies under /root/reference.
// TEST INFRASTRUCTURE ONLY.  PerRayRadiance=false is the configuration the 3DGUT renderer instantiates
// (per-particle radiance precomputed by the projection, gutKBufferRenderer.cuh:642-716).
#include "shim/gut_shim.h"
#define PARTICLE_RADIANCE_NUM_COEFFS 16
#define GAUSSIAN_PARTICLE_MAX_ALPHA 0.99f
#include <3dgut/kernels/cuda/models/gaussianParticles.cuh>

extern "C" {

int ref_gut_degree(void) { return PARTICLE_KERNEL_DEGREE; }

// processHitFwd<DEG,false,false>: feat3 = per-particle radiance; state5 = {T, rad[3], depth}
int ref_gut_process_hit_fwd(const float* ray_o, const float* ray_d, const float* density12, const float* feat3, float min_response,
                            float min_alpha, float* state5) {
    threedgut::ParticleDensity p;
    std::memcpy(&p, density12, sizeof(p));
    float T = state5[0], depth = state5[4];
    float3 rad = make_float3(state5[1], state5[2], state5[3]);
    const bool acc = threedgut::processHitFwd<PARTICLE_KERNEL_DEGREE, false, false>(
        make_float3(ray_o[0], ray_o[1], ray_o[2]), make_float3(ray_d[0], ray_d[1], ray_d[2]), 0, &p, feat3, min_response, min_alpha, 0, &T,
        &rad, &depth, nullptr);
    state5[0] = T; state5[1] = rad.x; state5[2] = rad.y; state5[3] = rad.z; state5[4] = depth;
    return acc ? 1 : 0;
}

void ref_gut_process_hit_bwd(const float* ray_o, const float* ray_d, const float* density12, const float* feat3, float min_response,
                             float min_alpha, float min_transmittance, float* state5, const float* fin5, const float* grads5,
                             float* g_density12, float* g_feat3) {
    threedgut::ParticleDensity p, g;
    std::memcpy(&p, density12, sizeof(p));
    std::memset(&g, 0, sizeof(g));
    g_feat3[0] = g_feat3[1] = g_feat3[2] = 0.f;
    float T = state5[0], depth = state5[4];
    float3 rad = make_float3(state5[1], state5[2], state5[3]);
    threedgut::processHitBwd<PARTICLE_KERNEL_DEGREE, false, false>(
        make_float3(ray_o[0], ray_o[1], ray_o[2]), make_float3(ray_d[0], ray_d[1], ray_d[2]), 0, p, &g, feat3, g_feat3, min_response,
        min_alpha, min_transmittance, 0, fin5[0], T, grads5[0], make_float3(fin5[1], fin5[2], fin5[3]), rad,
        make_float3(grads5[1], grads5[2], grads5[3]), fin5[4], depth, grads5[4]);
    state5[0] = T; state5[1] = rad.x; state5[2] = rad.y; state5[3] = rad.z; state5[4] = depth;
    std::memcpy(g_density12, &g, sizeof(g));
}

void ref_gut_radiance_from_sph(int deg, const float* sph48, const float* dir3, float* out3) {
    const float3 r = threedgut::radianceFromSpH(deg, reinterpret_cast<const float3*>(sph48), make_float3(dir3[0], dir3[1], dir3[2]));
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}

}  // extern "C"

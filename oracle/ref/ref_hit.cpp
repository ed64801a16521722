// ref_hit.cpp — C entry points over the REFERENCE's own per-hit math, compiled on the host from the sources where
// they lie under /root/reference (threedgrt_tracer/include/3dgrt/kernels/cuda/gaussianParticles.cuh).
// TEST INFRASTRUCTURE ONLY: validates oracle/gut_oracle.c / grt_oracle.c and generates tests/golden/*.npz.
// Built once per particle kernel degree (-DPARTICLE_KERNEL_DEGREE=2|4) into oracle/_ref/libref_hit_degN.so.
#include "shim/cuda_shim.h"
#define SPH_MAX_NUM_COEFFS 16
#define GAUSSIAN_PARTICLE_MAX_ALPHA 0.99f
#include <3dgrt/kernels/cuda/gaussianParticles.cuh>

extern "C" {

int ref_degree(void) { return PARTICLE_KERNEL_DEGREE; }

// processHit<DEG,false>: state = {T, rad[3], depth, normal[3]}; returns accept
int ref_process_hit(const float* ray_o, const float* ray_d, const float* density12, const float* sph48, float min_response,
                    float min_alpha, int sph_degree, int with_normal, float* state8) {
    ParticleDensity p;
    std::memcpy(&p, density12, sizeof(p));
    float T = state8[0], depth = state8[4];
    float3 rad = make_float3(state8[1], state8[2], state8[3]);
    float3 nrm = make_float3(state8[5], state8[6], state8[7]);
    const bool acc = processHit<PARTICLE_KERNEL_DEGREE, false>(make_float3(ray_o[0], ray_o[1], ray_o[2]),
                                                               make_float3(ray_d[0], ray_d[1], ray_d[2]), 0, &p, sph48, min_response,
                                                               min_alpha, sph_degree, &T, &rad, &depth, with_normal ? &nrm : nullptr);
    state8[0] = T; state8[1] = rad.x; state8[2] = rad.y; state8[3] = rad.z; state8[4] = depth;
    state8[5] = nrm.x; state8[6] = nrm.y; state8[7] = nrm.z;
    return acc ? 1 : 0;
}

// processHitBwd<DEG,false>: state5 = running {T, rad[3], depth} (in/out); fin5 = forward results; grads5 = {dL/dT, dL/drad[3], dL/ddepth}
void ref_process_hit_bwd(const float* ray_o, const float* ray_d, const float* density12, const float* sph48, float min_response,
                         float min_alpha, float min_transmittance, int sph_degree, float* state5, const float* fin5,
                         const float* grads5, float* g_density12, float* g_sph48) {
    ParticleDensity p, g;
    std::memcpy(&p, density12, sizeof(p));
    std::memset(&g, 0, sizeof(g));
    std::memset(g_sph48, 0, sizeof(float) * 48);
    float T = state5[0], depth = state5[4];
    float3 rad = make_float3(state5[1], state5[2], state5[3]);
    processHitBwd<PARTICLE_KERNEL_DEGREE, false>(make_float3(ray_o[0], ray_o[1], ray_o[2]), make_float3(ray_d[0], ray_d[1], ray_d[2]), 0, &p,
                                                 &g, sph48, g_sph48, min_response, min_alpha, min_transmittance, sph_degree, fin5[0], T,
                                                 grads5[0], make_float3(fin5[1], fin5[2], fin5[3]), rad,
                                                 make_float3(grads5[1], grads5[2], grads5[3]), fin5[4], depth, grads5[4]);
    state5[0] = T; state5[1] = rad.x; state5[2] = rad.y; state5[3] = rad.z; state5[4] = depth;
    std::memcpy(g_density12, &g, sizeof(g));
}

void ref_radiance_from_sph(int deg, const float* sph48, const float* dir3, float* out3) {
    const float3 r = radianceFromSpH(deg, reinterpret_cast<const float3*>(sph48), make_float3(dir3[0], dir3[1], dir3[2]));
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}

// radianceFromSpHBwd: returns the (clamped) radiance, accumulates weight*radGrad-weighted SH gradients into g_sph48 (zeroed here)
void ref_radiance_from_sph_bwd(int deg, const float* sph48, const float* dir3, float weight, const float* rad_grad3, float* out3,
                               float* g_sph48) {
    std::memset(g_sph48, 0, sizeof(float) * 48);
    const float3 r = radianceFromSpHBwd(deg, reinterpret_cast<const float3*>(sph48), make_float3(dir3[0], dir3[1], dir3[2]), weight,
                                        make_float3(rad_grad3[0], rad_grad3[1], rad_grad3[2]), reinterpret_cast<float3*>(g_sph48));
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}

int ref_intersect_custom(const float* ray_o, const float* ray_d, const float* density12, float tmin, float tmax, float max_sqdist,
                         float* hit_t) {
    ParticleDensity p;
    std::memcpy(&p, density12, sizeof(p));
    float t = 0.f;
    const bool ok = intersectCustomParticle(make_float3(ray_o[0], ray_o[1], ray_o[2]), make_float3(ray_d[0], ray_d[1], ray_d[2]), 0, &p, tmin,
                                            tmax, max_sqdist, t);
    *hit_t = t;
    return ok ? 1 : 0;
}

int ref_intersect_instance(const float* pray_o, const float* pray_d, float tmin, float tmax, float max_sqdist, float* hit_t) {
    float t = 0.f;
    const bool ok = intersectInstanceParticle(make_float3(pray_o[0], pray_o[1], pray_o[2]), make_float3(pray_d[0], pray_d[1], pray_d[2]), 0,
                                              tmin, tmax, max_sqdist, t);
    *hit_t = t;
    return ok ? 1 : 0;
}

float ref_particle_response(float gray) { return particleResponse<PARTICLE_KERNEL_DEGREE>(gray); }
float ref_particle_response_grd(float gray, float gres, float gres_grd) { return particleResponseGrd<PARTICLE_KERNEL_DEGREE>(gray, gres, gres_grd); }

}  // extern "C"

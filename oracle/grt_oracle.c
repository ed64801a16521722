/*
 * grt_oracle.c — CPU restatement of the reference 3DGRT path (threedgrt_tracer).
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker;
 * never imported by the product package.
 *
 * Parity pinning: the per-hit math (orc_grt_process_hit_fwd/bwd, candidate tests) is checked against the reference's own
 * header (threedgrt_tracer/include/3dgrt/kernels/cuda/gaussianParticles.cuh) compiled on the host (oracle/_ref) through
 * the golden vectors in tests/golden/.  OptiX (un-vendored, v7.5) is replaced by its documented semantics: the
 * candidate set of a trace is every instance whose transformed unit box the ray interval touches and whose intersection
 * program accepts; any-hit keeps the 16 smallest hit distances (referenceOptix.cu:210-248).  Where OptiX leaves the
 * order undefined (equal distances) the oracle orders by (distance, particle index).
 *
 * Sequence restated: optixTracer.cpp:616-890 (proxies), referenceOptix.cu:103-186 (forward ray generation),
 * referenceBwdOptix.cu:103-170 (backward), 3dgrt/kernels/cuda/gaussianParticles.cuh:337-731 (per-hit math).
 */
#include "orc_math.h"
#include "orc_nht.h"
#include "../include/grut_amd.h"

#include <stdio.h>
#include <stdlib.h>

#define GRT_INVALID 0xFFFFFFFFu
#define GRT_MAX_K 64

/* particlePrimitives.cu:27-51 kernelScale */
real orc_grt_kernel_scale(real density, real min_response, int clamping, real degree) {
    const real modulation = clamping ? density : 1;
    const real mr = r_min(min_response / modulation, R_(0.97));
    if (degree < 0) {
        const real k = r_fabs(degree);
        const real s = 1 / r_pow(3, k);
        return r_pow((1 / (r_log(mr) - 1) + 1) / s, 1 / k);
    }
    if (degree == 0) return ((1 - mr) / 3) / R_(-0.329630334487);
    const real a = R_(-4.5) / r_pow(3, degree);
    return r_pow(r_log(mr) / a, 1 / degree);
}

/* render.primitive_type (GrtConfig::primitive_type; optixTracer.cpp:176-201).  0 = instances; 1..4 = the closed convex triangle meshes of
 * particlePrimitives.cu:63-496, which OptiX traverses as built-in triangles with back faces culled (referenceOptix.cu:62): the particle is
 * offered at the distance at which the ray ENTERS its proxy, i.e. - in the proxy's own frame - the fixed polyhedron of orc_polyhedra.h.
 * A process-wide setting of this test library (the entry points that take a GrtConfig set it from cfg->primitive_type). */
#include "orc_polyhedra.h"
static int g_prim = 0;
/* custom primitives under the SLANG pipelines (neural harmonic features): particleDensityHitCustom (gaussianParticles.slang:489-523) reports
 * canonicalRayDistance - the UNSIGNED distance of the maximum-response point - where intersectCustomParticle (gaussianParticles.cuh:407-441)
 * reports it with the sign of the ray parameter: a particle whose maximum lies BEHIND the ray origin is a candidate of the Slang pipeline.
 * Set by the feature entry points, cleared by every other one. */
static int g_custom_abs = 0;
void orc_grt_set_primitive(int prim) { g_prim = prim; g_custom_abs = 0; }
/* GRUT_PRIM_CUSTOM (render.primitive_type custom): per particle {world box min, max, kernelScale^2, 0} - orc_grt_custom_boxes fills it, the
 * caller keeps it alive and registers it here before tracing (process-wide, like the primitive type). */
static const real* g_box8 = NULL;
void orc_grt_set_custom_boxes(const real* box8) { g_box8 = box8; }

/* computeGaussianEnclosingInstancesKernel, particlePrimitives.cu:543-610: instance transform [R diag(kscl) | mu] over a
 * unit box.  Emitted here as the INVERSE map (what traversal needs): inst = {W rows (9), mu (3)}, W = diag(1/kscl) R^T,
 * so that the object-space ray is o' = W (o - mu), d' = W d.  Also the world AABB of the box, the pruning slack
 * sqrt(2) * max(kscl) and the scene AABB (optixTracer.cpp:870-888). */
int orc_grt_proxies(const GrtConfig* cfg, uint32_t N, const real* positions, const real* rotations, const real* scales,
                    const real* densities, real* inst12, real* aabb6, real* slack, real* scene6) {
    for (int k = 0; k < 3; ++k) { scene6[k] = R_(3.0e38); scene6[3 + k] = R_(-3.0e38); }
    for (uint32_t i = 0; i < N; ++i) {
        const v4 q = {rotations[4 * i], rotations[4 * i + 1], rotations[4 * i + 2], rotations[4 * i + 3]};
        const m33 rotT = quat_wxyz_to_rotT(q); /* rows of R^T */
        const real ks = orc_grt_kernel_scale(densities[i], (real)cfg->particle_kernel_min_response, cfg->particle_kernel_density_clamping,
                                             (real)cfg->particle_kernel_degree);
        const real kscl[3] = {ks * scales[3 * i], ks * scales[3 * i + 1], ks * scales[3 * i + 2]};
        real* o = inst12 + 12 * (size_t)i;
        for (int r = 0; r < 3; ++r) {
            o[3 * r] = rotT.r[r].x / kscl[r]; o[3 * r + 1] = rotT.r[r].y / kscl[r]; o[3 * r + 2] = rotT.r[r].z / kscl[r];
        }
        o[9] = positions[3 * i]; o[10] = positions[3 * i + 1]; o[11] = positions[3 * i + 2];
        if (cfg->primitive_type == 8) {
            /* render.primitive_type sphere - computeGaussianEnclosingSphereKernel, particlePrimitives.cu:386-403: centre mu, radius
             * max(scale) * kernelScale.  The record maps into the frame scaled by the radius, W = diag(1 / r); the world box is the sphere's. */
            const real rad = r_max(scales[3 * i], r_max(scales[3 * i + 1], scales[3 * i + 2])) * ks;
            const real ir = 1 / rad;
            for (int k = 0; k < 9; ++k) o[k] = 0;
            o[0] = ir; o[4] = ir; o[8] = ir;
            real* b = aabb6 + 6 * (size_t)i;
            for (int k = 0; k < 3; ++k) { b[k] = o[9 + k] - rad; b[3 + k] = o[9 + k] + rad; }
            slack[i] = R_(1.41421356237) * rad;
            for (int k = 0; k < 3; ++k) { scene6[k] = r_min(scene6[k], b[k]); scene6[3 + k] = r_max(scene6[3 + k], b[3 + k]); }
            continue;
        }
        /* world half extent_c = sum_r |R_cr| kscl_r with R_cr = rotT.r[r].c */
        const orc_polyhedron* ph = &orc_polyhedra[cfg->primitive_type];   /* (instances: ext = 1) */
        const real ex[3] = {kscl[0] * (real)ph->ext[0], kscl[1] * (real)ph->ext[1], kscl[2] * (real)ph->ext[2]};
        const real hx = r_fabs(rotT.r[0].x) * ex[0] + r_fabs(rotT.r[1].x) * ex[1] + r_fabs(rotT.r[2].x) * ex[2];
        const real hy = r_fabs(rotT.r[0].y) * ex[0] + r_fabs(rotT.r[1].y) * ex[1] + r_fabs(rotT.r[2].y) * ex[2];
        const real hz = r_fabs(rotT.r[0].z) * ex[0] + r_fabs(rotT.r[1].z) * ex[1] + r_fabs(rotT.r[2].z) * ex[2];
        real* b = aabb6 + 6 * (size_t)i;
        b[0] = o[9] - hx; b[1] = o[10] - hy; b[2] = o[11] - hz; b[3] = o[9] + hx; b[4] = o[10] + hy; b[5] = o[11] + hz;
        slack[i] = R_(1.41421356237) * r_max(ex[0], r_max(ex[1], ex[2]));
        for (int k = 0; k < 3; ++k) { scene6[k] = r_min(scene6[k], b[k]); scene6[3 + k] = r_max(scene6[3 + k], b[3 + k]); }
    }
    return 0;
}

/* computeGaussianEnclosingAABBKernel, particlePrimitives.cu:498-541: the bounding box of the 8 corners R (c * kscl) + mu, c = +-1.  The
 * extreme corner of an axis has all three products of one sign: mu -+ ((|R_c0| k0 + |R_c1| k1) + |R_c2| k2), in the kernel's own
 * left-to-right summation (the same expression as grt_proxy_kernel's). */
int orc_grt_custom_boxes(const GrtConfig* cfg, uint32_t N, const real* positions, const real* rotations, const real* scales, const real* densities,
                         real* box8) {
    for (uint32_t i = 0; i < N; ++i) {
        const v4 q = {rotations[4 * i], rotations[4 * i + 1], rotations[4 * i + 2], rotations[4 * i + 3]};
        const m33 rotT = quat_wxyz_to_rotT(q);
        const real ks = orc_grt_kernel_scale(densities[i], (real)cfg->particle_kernel_min_response, cfg->particle_kernel_density_clamping,
                                             (real)cfg->particle_kernel_degree);
        const real k0 = ks * scales[3 * i], k1 = ks * scales[3 * i + 1], k2 = ks * scales[3 * i + 2];
        const real ux = (r_fabs(rotT.r[0].x) * k0 + r_fabs(rotT.r[1].x) * k1) + r_fabs(rotT.r[2].x) * k2;
        const real uy = (r_fabs(rotT.r[0].y) * k0 + r_fabs(rotT.r[1].y) * k1) + r_fabs(rotT.r[2].y) * k2;
        const real uz = (r_fabs(rotT.r[0].z) * k0 + r_fabs(rotT.r[1].z) * k1) + r_fabs(rotT.r[2].z) * k2;
        real* b = box8 + 8 * (size_t)i;
        b[0] = positions[3 * i] - ux; b[1] = positions[3 * i + 1] - uy; b[2] = positions[3 * i + 2] - uz;
        b[3] = positions[3 * i] + ux; b[4] = positions[3 * i + 1] + uy; b[5] = positions[3 * i + 2] + uz;
        b[6] = ks * ks; b[7] = 0;
    }
    return 0;
}

/* ---- candidate test -------------------------------------------------------------------------
 * Instance traversal: transform the ray with the instance's inverse map, slab-test the unit box [-1,1]^3 over the
 * current interval, then intersectInstanceParticle (gaussianParticles.cuh:449-466).  All in one arithmetic type.
 * The HIP kernel (grt_kernels.hip: candidate_abe) evaluates EXACTLY these operations in this order — hit order is compared
 * bit for bit — so the sequence is written out: fused multiply-adds where the device code has them (the reference's own
 * device code is compiled with nvcc's default contraction, its box test runs in the RT cores: neither fixes a rounding),
 * one correctly rounded division for the distance and one reciprocal per axis for the slabs, IEEE minNum / maxNum, and
 * the 3-sigma test |pd x po|^2 < 9 |pd|^4 without the normalisation (the same inequality as
 * |normalize(pd) x po|^2 / |pd|^2 < 9). */
typedef struct { real t, tnear, tfar; int ok; } grt_cand;


static real safe_rcp(real v) { return r_fabs(v) > R_(1e-30) ? 1 / v : (v < 0 || (v == 0 && 1 / v < 0) ? R_(-1.0e30) : R_(1.0e30)); }   /* grt_kernels.hip: safe_rcp */
static grt_cand candidate(const real* inst, v3 o, v3 d, real max_sqdist, uint32_t id) {
    grt_cand c; c.ok = 0; c.t = 0; c.tnear = 0; c.tfar = 0;
    const v3 dl = v3_make(o.x - inst[9], o.y - inst[10], o.z - inst[11]);
    const v3 po = v3_make(inst[0] * dl.x + inst[1] * dl.y + inst[2] * dl.z, inst[3] * dl.x + inst[4] * dl.y + inst[5] * dl.z,
                          inst[6] * dl.x + inst[7] * dl.y + inst[8] * dl.z);
    const v3 pd = v3_make(r_fma(inst[2], d.z, r_fma(inst[1], d.y, inst[0] * d.x)), r_fma(inst[5], d.z, r_fma(inst[4], d.y, inst[3] * d.x)),
                          r_fma(inst[8], d.z, r_fma(inst[7], d.y, inst[6] * d.x)));
    if (g_prim >= 1 && g_prim <= 4) {   /* closed triangle-mesh proxy: clip against the polyhedron's face planes, operation by operation as candidate_abe (grt_kernels.hip) */
        const orc_polyhedron* ph = &orc_polyhedra[g_prim];
        real tin = R_(-3.0e38), tout = R_(3.0e38);
        int miss = 0;
        for (int f = 0; f < ph->num_planes; ++f) {
            const real nx = (real)ph->planes[f][0], ny = (real)ph->planes[f][1], nz = (real)ph->planes[f][2], hh = (real)ph->planes[f][3];
            const real dn = r_fma(nz, pd.z, r_fma(ny, pd.y, nx * pd.x));
            const real on = hh - r_fma(nz, po.z, r_fma(ny, po.y, nx * po.x));
            const real tf = on * (1 / dn);   /* (round 6: reciprocal + product, as candidate_abe - one division per antipodal pair there) */
            if (dn < 0) tin = r_fmax(tin, tf);
            else if (dn > 0) tout = r_fmin(tout, tf);
            else if (on < 0) miss = 1;
        }
        c.t = tin; c.tnear = tin; c.tfar = R_(3.0e38);
        c.ok = !miss && (tin <= tout) && (tin > R_(-3.0e38));
        return c;
    }
    if (g_prim == 6) {   /* trisurfel (particlePrimitives.cu:155-205): the rhombus |x| + |y| <= sqrt 2 in the proxy's z = 0 plane, two triangles traced
                          * WITHOUT face culling (referenceOptix.cu:62: SurfelPrimitive -> OPTIX_RAY_FLAG_NONE); the reported distance is the plane's */
        if (pd.z == 0) return c;
        const real t = -po.z / pd.z;
        const real hx = r_fma(t, pd.x, po.x), hy = r_fma(t, pd.y, po.y);
        c.t = t; c.tnear = t; c.tfar = R_(3.0e38);
        c.ok = (r_fabs(hx) + r_fabs(hy) <= R_(1.4142135381698608));
        return c;
    }
    if (g_prim == 5) {   /* custom primitives: world box + the world-space intersection program (gaussianParticles.cuh:407-441); see candidate_abe */
        const real* bx = g_box8 + 8 * (size_t)id;
        const real jx = safe_rcp(d.x), jy = safe_rcp(d.y), jz = safe_rcp(d.z);
        const real ax0 = (bx[0] - o.x) * jx, ax1 = (bx[3] - o.x) * jx, ay0 = (bx[1] - o.y) * jy, ay1 = (bx[4] - o.y) * jy;
        const real az0 = (bx[2] - o.z) * jz, az1 = (bx[5] - o.z) * jz;
        const real tnear = r_fmax(r_fmax(r_fmin(ax0, ax1), r_fmin(ay0, ay1)), r_fmin(az0, az1));
        const real tfar  = r_fmin(r_fmin(r_fmax(ax0, ax1), r_fmax(ay0, ay1)), r_fmax(az0, az1));
        if (!(tnear <= tfar)) return c;
        c.tnear = tnear; c.tfar = tfar;
        const real numerator = -r_fma(po.z, pd.z, r_fma(po.y, pd.y, po.x * pd.x));
        const real dd = r_fma(pd.z, pd.z, r_fma(pd.y, pd.y, pd.x * pd.x));
        c.t = numerator / dd;
        if (g_custom_abs) c.t = r_fabs(c.t);
        const v3 cr = v3_make(r_fma(pd.y, po.z, -(pd.z * po.y)), r_fma(pd.z, po.x, -(pd.x * po.z)), r_fma(pd.x, po.y, -(pd.y * po.x)));
        c.ok = (r_fma(cr.z, cr.z, r_fma(cr.y, cr.y, cr.x * cr.x)) * bx[6] < max_sqdist * dd);
        return c;
    }
    /* slab test of the unit box: the reciprocals from one division where the product of the components is a comfortable normal number
     * (candidate_abe, round 6), one division per axis otherwise */
    real ax0, ax1, ay0, ay1, az0, az1;
    {
        const real pxy = pd.x * pd.y, prod = pxy * pd.z, aprod = r_fabs(prod);
        if (aprod >= R_(1e-24) && aprod <= R_(1e24)) {
            const real q = 1 / prod;
            const real ix = q * (pd.y * pd.z), iy = q * (pd.x * pd.z), iz = q * pxy;
            ax0 = r_fma(-po.x, ix, -ix); ax1 = r_fma(-po.x, ix, ix); ay0 = r_fma(-po.y, iy, -iy); ay1 = r_fma(-po.y, iy, iy);
            az0 = r_fma(-po.z, iz, -iz); az1 = r_fma(-po.z, iz, iz);
        } else {
            const real ix = 1 / pd.x, iy = 1 / pd.y, iz = 1 / pd.z;
            ax0 = (-1 - po.x) * ix; ax1 = (1 - po.x) * ix;
            ay0 = (-1 - po.y) * iy; ay1 = (1 - po.y) * iy;
            az0 = (-1 - po.z) * iz; az1 = (1 - po.z) * iz;
        }
    }
    const real tnear = r_fmax(r_fmax(r_fmin(ax0, ax1), r_fmin(ay0, ay1)), r_fmin(az0, az1));
    const real tfar  = r_fmin(r_fmin(r_fmax(ax0, ax1), r_fmax(ay0, ay1)), r_fmax(az0, az1));
    if (!(tnear <= tfar)) return c;
    c.tnear = tnear; c.tfar = tfar;
    /* intersectInstanceParticle */
    const real numerator = -r_fma(po.z, pd.z, r_fma(po.y, pd.y, po.x * pd.x));
    const real dd = r_fma(pd.z, pd.z, r_fma(pd.y, pd.y, pd.x * pd.x));
    c.t = numerator / dd;
    const v3 cr = v3_make(r_fma(pd.y, po.z, -(pd.z * po.y)), r_fma(pd.z, po.x, -(pd.x * po.z)), r_fma(pd.x, po.y, -(pd.y * po.x)));
    c.ok = (r_fma(cr.z, cr.z, r_fma(cr.y, cr.y, cr.x * cr.x)) < max_sqdist * (dd * dd));
    return c;
}

/* ---- per-hit math: gaussianParticles.cuh:337-405 (processHit), :468-731 (processHitBwd) ------ */
typedef struct { v3 pos, scl; v4 quat; m33 rotT; real density; } grt_particle;
static grt_particle load_particle(const real* pd) {
    grt_particle p;
    p.pos = v3_make(pd[0], pd[1], pd[2]); p.density = pd[3];
    p.quat.x = pd[4]; p.quat.y = pd[5]; p.quat.z = pd[6]; p.quat.w = pd[7];
    p.scl = v3_make(pd[8], pd[9], pd[10]);
    p.rotT = quat_wxyz_to_rotT(p.quat);
    return p;
}
static v3 v3_max0(v3 a) { return v3_make(r_max(a.x, 0), r_max(a.y, 0), r_max(a.z, 0)); }

typedef struct { real T; v3 rad; real depth; v3 normal; } grt_ray_state;

static int process_hit(const GrtConfig* cfg, v3 ro, v3 rd, const real* pd, const real* sph, int sph_deg, grt_ray_state* s, int with_normal) {
    const grt_particle p = load_particle(pd);
    const v3 giscl = v3_make(1 / p.scl.x, 1 / p.scl.y, 1 / p.scl.z);
    const v3 gposc = v3_sub(ro, p.pos);
    const v3 gposcr = v3_mul_rows(gposc, &p.rotT);
    const v3 gro = v3_mul(giscl, gposcr);
    const v3 rdr = v3_mul_rows(rd, &p.rotT);
    const v3 grdu = v3_mul(giscl, rdr);
    const v3 grd = v3_safe_normalize(grdu);
    const int surfel = cfg->primitive_type == 6;   /* PipelineParameters::SurfelPrimitive: the hit is the ray's crossing of the particle's z = 0 plane */
    const v3 gcrod = surfel ? v3_add(gro, v3_make(grd.x * -gro.z / grd.z, grd.y * -gro.z / grd.z, grd.z * -gro.z / grd.z)) : v3_cross(grd, gro);
    const real gray = v3_dot(gcrod, gcrod);
    const real gres = particle_response(cfg->particle_kernel_degree, gray);
    const real galpha = r_min((real)cfg->particle_kernel_max_alpha, gres * p.density);
    const int accept = (gres > (real)cfg->particle_kernel_min_response) && (galpha > (real)cfg->particle_kernel_min_alpha);
    if (accept) {
        const real weight = galpha * s->T;
        const real pdot = surfel ? -gro.z / grd.z : v3_dot(grd, v3_scale(gro, -1));
        const v3 grds = v3_mul(p.scl, v3_scale(grd, pdot));
        const real hitT = r_sqrt(v3_dot(grds, grds));
        const v3 grad = v3_max0(sh_radiance_unclamped(sph_deg, sph, rd));
        s->rad = v3_add(s->rad, v3_scale(grad, weight));
        s->T *= (1 - galpha);
        s->depth += hitT * weight;
        if (with_normal) { /* :398-402 */
            const v3 psr = m33_mul_cols(&p.rotT, p.scl);
            if (surfel) {
                s->normal = v3_add(s->normal, v3_scale(v3_make(0, 0, (grd.z > 0 ? 1 : -1) * psr.z), weight));
            } else {
                const v3 q = v3_add(gro, v3_scale(grd, pdot - r_sqrt(9 - gray)));
                const v3 n = v3_safe_normalize(v3_mul(q, psr));
                s->normal = v3_add(s->normal, v3_scale(n, weight));
            }
        }
    }
    return accept;
}

typedef struct {
    real T; v3 rad; real depth;          /* running */
    real T_fin; v3 rad_fin; real depth_fin;
    real T_grad; v3 rad_grad; real depth_grad;
} grt_bwd_state;

/* per-hit gradients are ADDED into g_density12[12] and g_sph[3*ncoef] (the reference's atomicAdd targets) */
static void process_hit_bwd(const GrtConfig* cfg, v3 ro, v3 rd, const real* pd, const real* sph, int sph_deg, real min_T,
                            grt_bwd_state* r, real* g_density12, real* g_sph) {
    const grt_particle p = load_particle(pd);
    const v3 gscl = p.scl;
    const v3 giscl = v3_make(1 / gscl.x, 1 / gscl.y, 1 / gscl.z);
    const v3 gposc = v3_sub(ro, p.pos);
    const v3 gposcr = v3_mul_rows(gposc, &p.rotT);
    const v3 gro = v3_mul(giscl, gposcr);
    const v3 rdr = v3_mul_rows(rd, &p.rotT);
    const v3 grdu = v3_mul(giscl, rdr);
    const v3 grd = v3_safe_normalize(grdu);
    const int surfel = cfg->primitive_type == 6;   /* the SurfelPrimitive branches of gaussianParticles.cuh:512-521, :558-565, :628-659 */
    const v3 gcrod = surfel ? v3_add(gro, v3_make(grd.x * -gro.z / grd.z, grd.y * -gro.z / grd.z, grd.z * -gro.z / grd.z)) : v3_cross(grd, gro);
    const real gray = v3_dot(gcrod, gcrod);
    const real gres = particle_response(cfg->particle_kernel_degree, gray);
    const real galpha = r_min((real)cfg->particle_kernel_max_alpha, gres * p.density);
    if (!((gres > (real)cfg->particle_kernel_min_response) && (galpha > (real)cfg->particle_kernel_min_alpha))) return;

    const real pdot = surfel ? -gro.z / grd.z : v3_dot(grd, v3_scale(gro, -1));
    const v3 grdd = v3_scale(grd, pdot);
    const v3 grds = v3_mul(gscl, grdd);
    const real gsq = v3_dot(grds, grds);
    const real gdist = r_sqrt(gsq);
    const real T = r->T;
    const real weight = galpha * T;
    const real nextT = (1 - galpha) * T;

    r->depth += weight * gdist;
    const real resHitT = r_max(nextT <= min_T ? 0 : (r->depth_fin - r->depth) / nextT, 0);
    const real galphaRayHitGrd = (gdist - resHitT) * T * r->depth_grad;
    const v3 grdsRayHitGrd = gsq > 0 ? v3_scale(grds, (2 * weight) / (2 * gdist) * r->depth_grad) : v3_make(0, 0, 0);
    const v3 gsclRayHitGrd = v3_mul(grdd, grdsRayHitGrd);
    const real grdScaledDot = v3_dot(v3_mul(grdsRayHitGrd, gscl), grd);
    v3 grdRayHitGrd, groRayHitGrd;
    if (surfel) {   /* :558-561 */
        grdRayHitGrd = v3_sub(v3_scale(v3_mul(gscl, grdsRayHitGrd), pdot), v3_make(0, 0, (pdot / grd.z) * grdScaledDot));
        groRayHitGrd = v3_make(0, 0, -grdScaledDot / grd.z);
    } else {
        grdRayHitGrd = v3_sub(v3_scale(v3_mul(gscl, grdsRayHitGrd), pdot), v3_scale(gro, grdScaledDot));
        groRayHitGrd = v3_scale(grd, -grdScaledDot);
    }

    const real resTrm = galpha < R_(0.999999) ? r->T_fin / (1 - galpha) : T;
    const real galphaRayDnsGrd = resTrm * -r->T_grad;

    /* radianceFromSpHBwd :101-177: clamped radiance + SH coefficient gradients (clamp-masked) */
    const v3 gradu = sh_radiance_unclamped(sph_deg, sph, rd);
    const v3 grad = v3_max0(gradu);
    v3 dL = v3_scale(r->rad_grad, weight);
    if (!(gradu.x > 0)) dL.x = 0;
    if (!(gradu.y > 0)) dL.y = 0;
    if (!(gradu.z > 0)) dL.z = 0;
    real b[16];
    sh_basis(sph_deg, rd, b);
    const int nact = (sph_deg + 1) * (sph_deg + 1);
    for (int k = 0; k < nact; ++k) { g_sph[3 * k] += b[k] * dL.x; g_sph[3 * k + 1] += b[k] * dL.y; g_sph[3 * k + 2] += b[k] * dL.z; }

    r->rad = v3_add(r->rad, v3_scale(grad, weight));
    v3 resRad = v3_make(0, 0, 0);
    if (!(nextT <= min_T)) resRad = v3_max0(v3_scale(v3_sub(r->rad_fin, r->rad), 1 / nextT));
    const real common = galphaRayHitGrd + galphaRayDnsGrd + T * (grad.x - resRad.x) * r->rad_grad.x + T * (grad.y - resRad.y) * r->rad_grad.y +
                        T * (grad.z - resRad.z) * r->rad_grad.z;
    g_density12[3] += gres * common;
    const real gresGrd = p.density * common;
    const real grayGrd = particle_response_grd(cfg->particle_kernel_degree, gray, gres, gresGrd);

    v3 grdGrd, groGrd;
    if (surfel) {   /* :628-659: grayDist = |gro + grd ghitT|^2, ghitT = -gro.z / grd.z */
        const real ghitT = -gro.z / grd.z;
        const v3 ghitPos = v3_add(gro, v3_scale(grd, ghitT));
        const v3 ghitPosGrd = v3_scale(ghitPos, 2 * grayGrd);
        groGrd = ghitPosGrd;
        grdGrd = v3_scale(ghitPosGrd, ghitT);
        const real ghitTGrd = grd.x * ghitPosGrd.x + grd.y * ghitPosGrd.y + grd.z * ghitPosGrd.z;
        groGrd.z += -ghitTGrd / grd.z;
        grdGrd.z += (gro.z * ghitTGrd) / (grd.z * grd.z);
    } else {
        const v3 gcrodGrd = v3_scale(gcrod, 2 * grayGrd);
        grdGrd = v3_make(gcrodGrd.z * gro.y - gcrodGrd.y * gro.z, gcrodGrd.x * gro.z - gcrodGrd.z * gro.x, gcrodGrd.y * gro.x - gcrodGrd.x * gro.y);
        groGrd = v3_make(gcrodGrd.y * grd.z - gcrodGrd.z * grd.y, gcrodGrd.z * grd.x - gcrodGrd.x * grd.z, gcrodGrd.x * grd.y - gcrodGrd.y * grd.x);
    }
    const v3 groTot = v3_add(groGrd, groRayHitGrd);
    const v3 gsclGrdGro = v3_mul(v3_make(-gposcr.x / (gscl.x * gscl.x), -gposcr.y / (gscl.y * gscl.y), -gposcr.z / (gscl.z * gscl.z)), groTot);
    const v3 gposcrGrd = v3_mul(giscl, groTot);
    const v3 gposcGrd = matmul_bw_vec(&p.rotT, gposcrGrd);
    const v4 grotGrdPoscr = matmul_bw_quat(gposc, gposcrGrd, p.quat);
    g_density12[0] += -gposcGrd.x; g_density12[1] += -gposcGrd.y; g_density12[2] += -gposcGrd.z;
    const v3 grduGrd = v3_safe_normalize_bw(grdu, v3_add(grdGrd, grdRayHitGrd));
    g_density12[8] += gsclRayHitGrd.x + gsclGrdGro.x + (-rdr.x / (gscl.x * gscl.x)) * grduGrd.x;
    g_density12[9] += gsclRayHitGrd.y + gsclGrdGro.y + (-rdr.y / (gscl.y * gscl.y)) * grduGrd.y;
    g_density12[10] += gsclRayHitGrd.z + gsclGrdGro.z + (-rdr.z / (gscl.z * gscl.z)) * grduGrd.z;
    const v3 rdrGrd = v3_mul(giscl, grduGrd);
    const v4 grotGrdRd = matmul_bw_quat(rd, rdrGrd, p.quat);
    g_density12[4] += grotGrdPoscr.x + grotGrdRd.x; g_density12[5] += grotGrdPoscr.y + grotGrdRd.y;
    g_density12[6] += grotGrdPoscr.z + grotGrdRd.z; g_density12[7] += grotGrdPoscr.w + grotGrdRd.w;
    r->T = nextT;
}

/* ---- known-answer entry points (tests/golden/per_hit_deg*.npz: grt_* arrays) ------------------ */
static GrtConfig kat_config(int degree, real min_response, real min_alpha, real max_alpha) {
    GrtConfig cfg; memset(&cfg, 0, sizeof(cfg));
    cfg.particle_kernel_degree = degree; cfg.particle_kernel_min_response = (float)min_response;
    cfg.particle_kernel_min_alpha = (float)min_alpha; cfg.particle_kernel_max_alpha = (float)max_alpha;
    return cfg;
}
/* state8 = {T, rad[3], depth, normal[3]} */
int orc_grt_process_hit_fwd(int degree, real min_response, real min_alpha, real max_alpha, const real* ray_o, const real* ray_d,
                            const real* density12, const real* sph48, int sph_deg, int with_normal, real* state8) {
    const GrtConfig cfg = kat_config(degree, min_response, min_alpha, max_alpha);
    grt_ray_state s;
    s.T = state8[0]; s.rad = v3_make(state8[1], state8[2], state8[3]); s.depth = state8[4];
    s.normal = v3_make(state8[5], state8[6], state8[7]);
    const int acc = process_hit(&cfg, v3_make(ray_o[0], ray_o[1], ray_o[2]), v3_make(ray_d[0], ray_d[1], ray_d[2]), density12, sph48, sph_deg, &s, with_normal);
    state8[0] = s.T; state8[1] = s.rad.x; state8[2] = s.rad.y; state8[3] = s.rad.z; state8[4] = s.depth;
    state8[5] = s.normal.x; state8[6] = s.normal.y; state8[7] = s.normal.z;
    return acc;
}
/* ... for render.primitive_type `prim` (6 = trisurfel: the SurfelPrimitive branches of processHit); the tests composite a ray's hit
 * sequence hit by hit with it, in any order they choose (identification of order ties against the reference programs' goldens) */
int orc_grt_process_hit_fwd_prim(int prim, int degree, real min_response, real min_alpha, real max_alpha, const real* ray_o, const real* ray_d,
                                 const real* density12, const real* sph48, int sph_deg, real* state8) {
    GrtConfig cfg = kat_config(degree, min_response, min_alpha, max_alpha);
    cfg.primitive_type = prim;
    grt_ray_state s;
    s.T = state8[0]; s.rad = v3_make(state8[1], state8[2], state8[3]); s.depth = state8[4];
    s.normal = v3_make(state8[5], state8[6], state8[7]);
    const int acc = process_hit(&cfg, v3_make(ray_o[0], ray_o[1], ray_o[2]), v3_make(ray_d[0], ray_d[1], ray_d[2]), density12, sph48, sph_deg, &s, 0);
    state8[0] = s.T; state8[1] = s.rad.x; state8[2] = s.rad.y; state8[3] = s.rad.z; state8[4] = s.depth;
    return acc;
}
void orc_grt_process_hit_bwd(int degree, real min_response, real min_alpha, real max_alpha, real min_transmittance, const real* ray_o,
                             const real* ray_d, const real* density12, const real* sph48, int sph_deg, real* state5, const real* fin5,
                             const real* grads5, real* g_density12, real* g_sph48) {
    const GrtConfig cfg = kat_config(degree, min_response, min_alpha, max_alpha);
    grt_bwd_state r;
    r.T = state5[0]; r.rad = v3_make(state5[1], state5[2], state5[3]); r.depth = state5[4];
    r.T_fin = fin5[0]; r.rad_fin = v3_make(fin5[1], fin5[2], fin5[3]); r.depth_fin = fin5[4];
    r.T_grad = grads5[0]; r.rad_grad = v3_make(grads5[1], grads5[2], grads5[3]); r.depth_grad = grads5[4];
    for (int k = 0; k < 12; ++k) g_density12[k] = 0;
    for (int k = 0; k < 48; ++k) g_sph48[k] = 0;
    process_hit_bwd(&cfg, v3_make(ray_o[0], ray_o[1], ray_o[2]), v3_make(ray_d[0], ray_d[1], ray_d[2]), density12, sph48, sph_deg,
                    min_transmittance, &r, g_density12, g_sph48);
    state5[0] = r.T; state5[1] = r.rad.x; state5[2] = r.rad.y; state5[3] = r.rad.z; state5[4] = r.depth;
}
/* intersectInstanceParticle on an object-space ray */
int orc_grt_intersect_instance(const real* pray_o, const real* pray_d, real tmin, real tmax, real max_sqdist, real* hit_t) {
    const v3 pd = v3_make(pray_d[0], pray_d[1], pray_d[2]), po = v3_make(pray_o[0], pray_o[1], pray_o[2]);
    const real dd = v3_dot(pd, pd);
    const real denominator = 1 / dd;
    const real t = -(v3_dot(po, pd)) * denominator;
    *hit_t = t;
    if (!((t > tmin) && (t < tmax))) return 0;
    const v3 n = dd > 0 ? v3_scale(pd, 1 / r_sqrt(dd)) : pd;
    const v3 cr = v3_cross(n, po);
    return v3_dot(cr, cr) * denominator < max_sqdist;
}

/* ---- rays ------------------------------------------------------------------------------------ */
static v3 xform_point(const real* m12, v3 p) { /* pipelineParameters.h:97-105, row-major 3x4 */
    return v3_make(m12[0] * p.x + m12[1] * p.y + m12[2] * p.z + m12[3], m12[4] * p.x + m12[5] * p.y + m12[6] * p.z + m12[7],
                   m12[8] * p.x + m12[9] * p.y + m12[10] * p.z + m12[11]);
}
static v3 xform_dir(const real* m12, v3 p) {
    return v3_make(m12[0] * p.x + m12[1] * p.y + m12[2] * p.z, m12[4] * p.x + m12[5] * p.y + m12[6] * p.z, m12[8] * p.x + m12[9] * p.y + m12[10] * p.z);
}
/* referenceOptix.cu:33-39 intersectAABB */
static void scene_interval(const real* aabb6, v3 o, v3 d, real* tmin_o, real* tmax_o) {
    const real t0x = (aabb6[0] - o.x) / d.x, t0y = (aabb6[1] - o.y) / d.y, t0z = (aabb6[2] - o.z) / d.z;
    const real t1x = (aabb6[3] - o.x) / d.x, t1y = (aabb6[4] - o.y) / d.y, t1z = (aabb6[5] - o.z) / d.z;
    const real mx = r_max(t0x, t1x), my = r_max(t0y, t1y), mz = r_max(t0z, t1z);
    const real nx = r_min(t0x, t1x), ny = r_min(t0y, t1y), nz = r_min(t0z, t1z);
    *tmin_o = r_max(0, r_max(nx, r_max(ny, nz)));
    *tmax_o = r_min(mx, r_min(my, mz));
}

typedef struct { real t; uint32_t id; real tnear, tfar; } grt_hit;
static int hit_cmp(const void* a, const void* b) {
    const grt_hit* x = (const grt_hit*)a; const grt_hit* y = (const grt_hit*)b;
    if (x->t != y->t) return x->t < y->t ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}
/* all candidates of one ray, sorted by (t, id); brute force over the particles */
/* Optional prefilter (test infrastructure for the 1 M-particle frames): the per-packet candidate lists the GPU built for its own forward -
 * `ranges[packet][2]` into `entries` (particle ids; the top bit is a flag of the GPU's, 0xFFFFFFFF pads), `ray_packet[ray]` = the 8x8 packet
 * of each ray.  A list holds every particle whose proxy box some ray of the packet can touch (conservative by construction), so
 * restricting the all-pairs scan to it must not change any ray's candidate set: tests/parity_util.grt_full_parity checks exactly that on the
 * rays it also runs through all pairs, then uses the prefilter to compare 15 x more rays. */
static const uint32_t *g_pre_ranges = NULL, *g_pre_entries = NULL, *g_pre_ray_packet = NULL;
void orc_grt_set_candidate_prefilter(const uint32_t* ranges, const uint32_t* entries, const uint32_t* ray_packet) {
    g_pre_ranges = ranges; g_pre_entries = entries; g_pre_ray_packet = ray_packet;
}
/* trihexa (particlePrimitives.cu:107-153; checker only): three rhombi |a| + |b| <= sqrt 2 in the proxy's coordinate planes, traced as triangles with
 * back faces culled.  The windings make the x = 0 rhombus face +x, the y = 0 rhombus +y, and the two triangles of the z = 0 rhombus face opposite
 * ways (the x >= 0 half +z, the x <= 0 half -z): a ray is offered the particle once per front-facing piece it crosses - up to three times, at
 * three distances - and the any-hit / processHit programs treat every offer as a hit of the particle. */
static uint32_t trihexa_candidates_of(const real* inst, v3 o, v3 d, uint32_t id, unsigned planes /* bit k: the rhombus of plane k is offered */, grt_hit* out) {
    const v3 dl = v3_make(o.x - inst[9], o.y - inst[10], o.z - inst[11]);
    const v3 po = v3_make(inst[0] * dl.x + inst[1] * dl.y + inst[2] * dl.z, inst[3] * dl.x + inst[4] * dl.y + inst[5] * dl.z,
                          inst[6] * dl.x + inst[7] * dl.y + inst[8] * dl.z);
    const v3 pd = v3_make(r_fma(inst[2], d.z, r_fma(inst[1], d.y, inst[0] * d.x)), r_fma(inst[5], d.z, r_fma(inst[4], d.y, inst[3] * d.x)),
                          r_fma(inst[8], d.z, r_fma(inst[7], d.y, inst[6] * d.x)));
    const real D = R_(1.4142135381698608);
    uint32_t n = 0;
    if ((planes & 1u) && pd.x < 0) {   /* the x = 0 rhombus, seen from +x */
        const real t = -po.x / pd.x, hy = r_fma(t, pd.y, po.y), hz = r_fma(t, pd.z, po.z);
        if (r_fabs(hy) + r_fabs(hz) <= D) { out[n].t = t; out[n].id = id; out[n].tnear = t; out[n].tfar = R_(3.0e38); n++; }
    }
    if ((planes & 2u) && pd.y < 0) {   /* the y = 0 rhombus, seen from +y */
        const real t = -po.y / pd.y, hx = r_fma(t, pd.x, po.x), hz = r_fma(t, pd.z, po.z);
        if (r_fabs(hx) + r_fabs(hz) <= D) { out[n].t = t; out[n].id = id; out[n].tnear = t; out[n].tfar = R_(3.0e38); n++; }
    }
    if ((planes & 4u) && pd.z != 0) {  /* the z = 0 rhombus: its x >= 0 half seen from +z, its x <= 0 half from -z */
        const real t = -po.z / pd.z, hx = r_fma(t, pd.x, po.x), hy = r_fma(t, pd.y, po.y);
        if (r_fabs(hx) + r_fabs(hy) <= D && ((hx > 0 && pd.z < 0) || (hx < 0 && pd.z > 0))) {
            out[n].t = t; out[n].id = id; out[n].tnear = t; out[n].tfar = R_(3.0e38); n++;
        }
    }
    return n;
}
static uint32_t trihexa_candidates(const real* inst, v3 o, v3 d, uint32_t id, grt_hit* out) { return trihexa_candidates_of(inst, o, d, id, 7u, out); }
/* sphere (optixTracer.cpp:189-190, 765-781, 823-833; checker for GRUT_PRIM_SPHERE): OptiX's built-in sphere intersector offers the any-hit
 * program the ray's ENTRY into the particle's enclosing sphere and, that offer being ignored (__anyhit__ah keeps only the one that fills its
 * payload), its EXIT: two offers per particle, at the two roots of |po + t pd|^2 = 1 in the frame scaled by the radius.  NVIDIA does not
 * publish the intersector's arithmetic; this sequence - the emulated OptiX's (oracle/ref/ref_grt_emul.inl) and candidate_abe's
 * (grt_kernels.hip), operation by operation - is the definition here. */
static uint32_t sphere_candidates(const real* inst, v3 o, v3 d, uint32_t id, grt_hit* out) {
    const v3 dl = v3_make(o.x - inst[9], o.y - inst[10], o.z - inst[11]);
    const v3 po = v3_make(inst[0] * dl.x + inst[1] * dl.y + inst[2] * dl.z, inst[3] * dl.x + inst[4] * dl.y + inst[5] * dl.z,
                          inst[6] * dl.x + inst[7] * dl.y + inst[8] * dl.z);
    const v3 pd = v3_make(r_fma(inst[2], d.z, r_fma(inst[1], d.y, inst[0] * d.x)), r_fma(inst[5], d.z, r_fma(inst[4], d.y, inst[3] * d.x)),
                          r_fma(inst[8], d.z, r_fma(inst[7], d.y, inst[6] * d.x)));
    const real qa = r_fma(pd.z, pd.z, r_fma(pd.y, pd.y, pd.x * pd.x)), qb = r_fma(po.z, pd.z, r_fma(po.y, pd.y, po.x * pd.x));
    const real qc = r_fma(po.z, po.z, r_fma(po.y, po.y, po.x * po.x)) - 1;
    const real disc = r_fma(qb, qb, -(qa * qc));
    if (!(disc >= 0) || !(qa > 0)) return 0;
    const real sq = r_sqrt(disc);
    out[0].t = (-qb - sq) / qa; out[0].id = id; out[0].tnear = out[0].t; out[0].tfar = R_(3.0e38);
    out[1].t = (-qb + sq) / qa; out[1].id = id; out[1].tnear = out[1].t; out[1].tfar = R_(3.0e38);
    return 2;
}
static uint32_t ray_candidates(uint32_t N, const real* inst12, v3 o, v3 d, grt_hit* out, uint32_t ray) {
    uint32_t n = 0;
    if (g_prim == 8) {   /* (two offers per particle) */
        if (g_pre_ranges && ray != 0xFFFFFFFFu) {   /* the GPU's lists hold PROXIES 2 i (entry) and 2 i + 1 (exit), binned by one box: the even one names the particle */
            const uint32_t p = g_pre_ray_packet[ray];
            for (uint32_t e = g_pre_ranges[2 * p]; e < g_pre_ranges[2 * p + 1]; ++e) {
                uint32_t i = g_pre_entries[e];
                if (i == 0xFFFFFFFFu) continue;
                i &= 0x7FFFFFFFu;
                if ((i & 1u) || (i >> 1) >= N) continue;
                n += sphere_candidates(inst12 + 12 * (size_t)(i >> 1), o, d, i >> 1, out + n);
            }
        } else {
            for (uint32_t i = 0; i < N; ++i) n += sphere_candidates(inst12 + 12 * (size_t)i, o, d, i, out + n);
        }
        qsort(out, n, sizeof(grt_hit), hit_cmp);
        return n;
    }
    if (g_prim == 7) {   /* (up to three offers per particle: the callers' buffers hold 3 N + 3 entries) */
        if (g_pre_ranges && ray != 0xFFFFFFFFu) {   /* the GPU's lists hold PROXIES 3 i + plane, every rhombus binned by its own box: one offer each */
            const uint32_t p = g_pre_ray_packet[ray];
            for (uint32_t e = g_pre_ranges[2 * p]; e < g_pre_ranges[2 * p + 1]; ++e) {
                uint32_t q = g_pre_entries[e];
                if (q == 0xFFFFFFFFu) continue;
                q &= 0x7FFFFFFFu;
                if (q / 3u >= N) continue;
                n += trihexa_candidates_of(inst12 + 12 * (size_t)(q / 3u), o, d, q / 3u, 1u << (q % 3u), out + n);
            }
            qsort(out, n, sizeof(grt_hit), hit_cmp);
            return n;
        }
        for (uint32_t i = 0; i < N; ++i) n += trihexa_candidates(inst12 + 12 * (size_t)i, o, d, i, out + n);
        qsort(out, n, sizeof(grt_hit), hit_cmp);
        return n;
    }
    if (g_pre_ranges && ray != 0xFFFFFFFFu) {
        const uint32_t p = g_pre_ray_packet[ray];
        for (uint32_t e = g_pre_ranges[2 * p]; e < g_pre_ranges[2 * p + 1]; ++e) {
            uint32_t i = g_pre_entries[e];
            if (i == 0xFFFFFFFFu) continue;
            i &= 0x7FFFFFFFu;
            if (i >= N) continue;
            const grt_cand c = candidate(inst12 + 12 * (size_t)i, o, d, R_(9.0), i);
            if (c.ok) { out[n].t = c.t; out[n].id = i; out[n].tnear = c.tnear; out[n].tfar = c.tfar; n++; }
        }
        qsort(out, n, sizeof(grt_hit), hit_cmp);
        return n;
    }
    for (uint32_t i = 0; i < N; ++i) {
        const grt_cand c = candidate(inst12 + 12 * (size_t)i, o, d, R_(9.0), i); /* hitMaxParticleSquaredDistance, pipelineParameters.h:71 */
        if (c.ok) { out[n].t = c.t; out[n].id = i; out[n].tnear = c.tnear; out[n].tfar = c.tfar; n++; }
    }
    qsort(out, n, sizeof(grt_hit), hit_cmp);
    return n;
}
/* one optixTrace: up to K nearest candidates with t in (tmin, tmax) whose box interval touches [tmin, tmax] */
static int trace_round(const grt_hit* cands, uint32_t n, real tmin, real tmax, int K, grt_hit* out) {
    int k = 0;
    for (uint32_t i = 0; i < n && k < K; ++i) {
        const grt_hit* h = &cands[i];
        if ((h->t > tmin) && (h->t < tmax) && (h->tfar >= tmin) && (h->tnear <= tmax)) out[k++] = *h;
    }
    return k;
}

/* render.pipeline_type barycentricSurfels - __raygen__rg / __anyhit__ah of barycentricSurfelsOptix.cu:84-228 (FORWARD ONLY in the reference).
 * Trisurfel proxies traced without face culling; a trace returns the TEN nearest triangle hits beyond the last hit distance, each with the
 * squared distance of the hit point from the surfel's centre in the proxy frame (computeTrisurfelSquaredDistance from the triangle's
 * barycentrics, :179-188: both triangles of a surfel map onto |(x, y)|^2 of the plane crossing); per hit: the kernel response SCALED to the
 * proxy frame (particleScaledResponse, gaussianParticles.cuh:296-333, with the density-modulated minimum response), alpha = min(0.99, response
 * x density), radiance from SH along the ray, depth += HIT distance x weight, normal = the surfel's, flipped along the ray (:160-164).
 * The plane crossing is evaluated as the trisurfel candidate test does (grt_kernels.hip: surfel_crossing, operation by operation). */
static real scaled_response(int degree, int clamped, real gray, real modulated_min_response, real modulation) {
    const real min_response = r_min(modulated_min_response / modulation, R_(0.97));
    const real lm = clamped ? r_log(min_response) : modulated_min_response;
    switch (degree) {
    case 8: { const real g2 = gray * gray; return r_exp(lm * g2 * g2); }
    case 5: return r_exp(lm * gray * gray * r_sqrt(gray));
    case 4: return r_exp(lm * gray * gray);
    case 3: return r_exp(lm * gray * r_sqrt(gray));
    case 1: return r_exp(lm * r_sqrt(gray));
    case 0: { const real s = (1 - min_response) / 3; return r_max(1 + s * r_sqrt(gray), 0); }
    default: return r_exp(lm * gray);
    }
}
typedef struct { real t, sq; uint32_t id; } bary_hit;
static int bary_cmp(const void* a, const void* b) {
    const bary_hit* x = (const bary_hit*)a; const bary_hit* y = (const bary_hit*)b;
    if (x->t != y->t) return x->t < y->t ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}
static int trace_bary_fwd(const GrtConfig* cfg, uint32_t N, const real* density12, const real* sph, int sph_deg, real min_T, const real* inst12,
                          const real* scene6, const real* ray_to_world12, uint32_t nrays, const real* ray_o, const real* ray_d, real* out_rad,
                          real* out_dns, real* out_hit2, real* out_nrm, real* out_cnt, int32_t* visibility, uint32_t* dbg_ids, uint32_t* dbg_count,
                          uint32_t dbg_cap) {
    const int K = 10;   /* MaxNumHitPerTrace, barycentricSurfelsOptix.cu:26 */
    const int ncoef = (cfg->particle_radiance_sph_degree + 1) * (cfg->particle_radiance_sph_degree + 1);
    const real eps = R_(1e-9);
    const int clamped = cfg->particle_kernel_density_clamping != 0, degree = cfg->particle_kernel_degree;
    const real hit_min = (real)cfg->particle_kernel_min_response;
    const real scale_min = (clamped || degree == 0) ? hit_min : r_log(hit_min);   /* particleScaleMinResponse, :108-109 */
#pragma omp parallel
    {
        bary_hit* cands = (bary_hit*)malloc(sizeof(bary_hit) * ((size_t)N + 1));
#pragma omp for schedule(dynamic, 8)
        for (uint32_t r = 0; r < nrays; ++r) {
            const v3 o = xform_point(ray_to_world12, v3_make(ray_o[3 * r], ray_o[3 * r + 1], ray_o[3 * r + 2]));
            const v3 d = xform_dir(ray_to_world12, v3_make(ray_d[3 * r], ray_d[3 * r + 1], ray_d[3 * r + 2]));
            uint32_t n = 0;
            for (uint32_t i = 0; i < N; ++i) {
                const real* inst = inst12 + 12 * (size_t)i;
                const v3 dl = v3_make(o.x - inst[9], o.y - inst[10], o.z - inst[11]);
                const v3 po = v3_make(inst[0] * dl.x + inst[1] * dl.y + inst[2] * dl.z, inst[3] * dl.x + inst[4] * dl.y + inst[5] * dl.z,
                                      inst[6] * dl.x + inst[7] * dl.y + inst[8] * dl.z);
                const v3 pd = v3_make(r_fma(inst[2], d.z, r_fma(inst[1], d.y, inst[0] * d.x)), r_fma(inst[5], d.z, r_fma(inst[4], d.y, inst[3] * d.x)),
                                      r_fma(inst[8], d.z, r_fma(inst[7], d.y, inst[6] * d.x)));
                if (pd.z == 0) continue;
                const real t = -po.z / pd.z;
                const real hx = r_fma(t, pd.x, po.x), hy = r_fma(t, pd.y, po.y);
                if (!(r_fabs(hx) + r_fabs(hy) <= R_(1.4142135381698608))) continue;
                cands[n].t = t; cands[n].id = i; cands[n].sq = r_fma(hy, hy, hx * hx); n++;
            }
            qsort(cands, n, sizeof(bary_hit), bary_cmp);
            real T = 1, depth = 0, cnt = 0;
            v3 rad = v3_make(0, 0, 0), nrm = v3_make(0, 0, 0);
            real tEnter, tExit;
            scene_interval(scene6, o, d, &tEnter, &tExit);
            real tLast = r_max(0, tEnter - eps);      /* intersectAABB, :38-44: {max(0, tnear - eps), tfar + eps} */
            const real tMax = tExit + eps;
            uint32_t ndbg = 0, at = 0;
            while ((tLast <= tMax) && (T > min_T)) {
                /* the ten nearest hits with t in (tLast + eps, tMax): the list is sorted, `at` only moves forward */
                while (at < n && !(cands[at].t > tLast + eps)) at++;
                int k = 0;
                for (; k < K && at + (uint32_t)k < n && cands[at + k].t < tMax; ++k) {}
                if (k == 0) break;
                for (int i = 0; i < k; ++i) {
                    const bary_hit* h = &cands[at + i];
                    if (T > min_T) {
                        const real* pdn = density12 + 12 * (size_t)h->id;
                        const real density = pdn[3];
                        const real response = scaled_response(degree, clamped, h->sq, scale_min, density);
                        const real alpha = r_min(R_(0.99), response * density);
                        if ((response > hit_min) && (alpha > (real)cfg->particle_kernel_min_alpha)) {
                            const real weight = alpha * T;
                            const v3 c = v3_max0(sh_radiance_unclamped(sph_deg, sph + (size_t)h->id * 3 * ncoef, d));
                            rad = v3_add(rad, v3_scale(c, weight));
                            T *= (1 - alpha);
                            depth += h->t * weight;
#pragma omp atomic write
                            visibility[h->id] = 1;
                            if (cfg->enable_normals) {   /* the surfel's normal as the trisurfel kernel leaves it: normalize(cross(v1 - v0, v2 - v0)) = -(third axis) */
                                const grt_particle p = load_particle(pdn);
                                const v3 n0 = v3_make(-p.rotT.r[2].x, -p.rotT.r[2].y, -p.rotT.r[2].z);
                                const real sgn = v3_dot(n0, d) < 0 ? R_(-1.0) : R_(1.0);
                                nrm = v3_add(nrm, v3_scale(n0, sgn * weight));
                            }
                            cnt += 1;
                        }
                        tLast = r_max(tLast, h->t);
                        if (dbg_ids && ndbg < dbg_cap) dbg_ids[(size_t)r * dbg_cap + ndbg] = h->id;
                        ndbg++;
                    }
                }
                at += (uint32_t)k;
            }
            out_rad[3 * r] = rad.x; out_rad[3 * r + 1] = rad.y; out_rad[3 * r + 2] = rad.z;
            out_dns[r] = 1 - T;
            out_hit2[2 * r] = depth; out_hit2[2 * r + 1] = tLast;
            if (cfg->enable_normals) { out_nrm[3 * r] = nrm.x; out_nrm[3 * r + 1] = nrm.y; out_nrm[3 * r + 2] = nrm.z; }
            if (cfg->enable_hitcounts) out_cnt[r] = cnt;
            if (dbg_count) dbg_count[r] = ndbg;
        }
        free(cands);
    }
    return 0;
}

/* __raygen__rg, referenceOptix.cu:103-186.  rays are [nrays,3] in ray space; outputs per ray.
 * dbg_ids (optional): [nrays, dbg_cap] processed candidates in order; dbg_count [nrays]. */
int orc_grt_trace_fwd(const GrtConfig* cfg, uint32_t N, const real* density12, const real* sph, int sph_deg, real min_T,
                      const real* inst12, const real* scene6, const real* ray_to_world12, uint32_t nrays, const real* ray_o,
                      const real* ray_d, real* out_rad, real* out_dns, real* out_hit2, real* out_nrm, real* out_cnt,
                      int32_t* visibility, uint32_t* dbg_ids, uint32_t* dbg_count, uint32_t dbg_cap) {
    orc_grt_set_primitive(cfg->primitive_type);
    if (cfg->pipeline_type == GRUT_PIPELINE_BARYCENTRIC_SURFELS)
        return cfg->primitive_type == 6 ? trace_bary_fwd(cfg, N, density12, sph, sph_deg, min_T, inst12, scene6, ray_to_world12, nrays, ray_o, ray_d, out_rad, out_dns,
                                                         out_hit2, out_nrm, out_cnt, visibility, dbg_ids, dbg_count, dbg_cap) : -2;
    const int K = cfg->max_hits_per_trace > 0 ? cfg->max_hits_per_trace : 16;
    if (K > GRT_MAX_K) return -1;
    const int ncoef = (cfg->particle_radiance_sph_degree + 1) * (cfg->particle_radiance_sph_degree + 1);
    const real eps = R_(1e-9);
#pragma omp parallel
    {
        grt_hit* cands = (grt_hit*)malloc(sizeof(grt_hit) * (3 * (size_t)N + 3))   /* (trihexa: up to three offers per particle) */;
#pragma omp for schedule(dynamic, 8)
        for (uint32_t r = 0; r < nrays; ++r) {
            const v3 o = xform_point(ray_to_world12, v3_make(ray_o[3 * r], ray_o[3 * r + 1], ray_o[3 * r + 2]));
            const v3 d = xform_dir(ray_to_world12, v3_make(ray_d[3 * r], ray_d[3 * r + 1], ray_d[3 * r + 2]));
            grt_ray_state s; s.T = 1; s.rad = v3_make(0, 0, 0); s.depth = 0; s.normal = v3_make(0, 0, 0);
            real cnt = 0;
            real tEnter, tExit;
            scene_interval(scene6, o, d, &tEnter, &tExit);
            real tLast = r_max(0, tEnter - eps);
            const uint32_t n = ray_candidates(N, inst12, o, d, cands, r);
            uint32_t ndbg = 0;
            grt_hit buf[GRT_MAX_K];
            while ((tLast <= tExit) && (s.T > min_T)) {
                const int k = trace_round(cands, n, tLast + eps, tExit + eps, K, buf);
                if (k == 0) break;
                for (int i = 0; i < k; ++i) {
                    if (s.T > min_T) {
                        const uint32_t id = buf[i].id;
                        const int acc = process_hit(cfg, o, d, density12 + 12 * (size_t)id, sph + (size_t)id * 3 * ncoef, sph_deg, &s, cfg->enable_normals);
                        if (acc) {
#pragma omp atomic write
                            visibility[id] = 1;
                        }
                        tLast = r_max(tLast, buf[i].t);
                        cnt += acc ? 1 : 0;
                        if (dbg_ids && ndbg < dbg_cap) dbg_ids[(size_t)r * dbg_cap + ndbg] = id;
                        ndbg++;
                    }
                }
            }
            out_rad[3 * r] = s.rad.x; out_rad[3 * r + 1] = s.rad.y; out_rad[3 * r + 2] = s.rad.z;
            out_dns[r] = 1 - s.T;
            out_hit2[2 * r] = s.depth; out_hit2[2 * r + 1] = tLast;
            if (cfg->enable_normals) { out_nrm[3 * r] = s.normal.x; out_nrm[3 * r + 1] = s.normal.y; out_nrm[3 * r + 2] = s.normal.z; }
            if (cfg->enable_hitcounts) out_cnt[r] = cnt;
            if (dbg_count) dbg_count[r] = ndbg;
        }
        free(cands);
    }
    return 0;
}

/* __raygen__rg of referenceBwdOptix.cu:103-170.  g_density12 [N,12], g_sph [N,3*ncoef] are accumulated into. */
int orc_grt_trace_bwd(const GrtConfig* cfg, uint32_t N, const real* density12, const real* sph, int sph_deg, real min_T,
                      const real* inst12, const real* scene6, const real* ray_to_world12, uint32_t nrays, const real* ray_o,
                      const real* ray_d, const real* rad, const real* dns, const real* hit2, const real* g_rad, const real* g_dns,
                      const real* g_hit, real* g_density12, real* g_sph, uint32_t* dbg_ids, uint32_t* dbg_count, uint32_t dbg_cap,
                      uint8_t* round_shift) {
    orc_grt_set_primitive(cfg->primitive_type);
    const int K = cfg->max_hits_per_trace > 0 ? cfg->max_hits_per_trace : 16;
    if (K > GRT_MAX_K) return -1;
    const int ncoef = (cfg->particle_radiance_sph_degree + 1) * (cfg->particle_radiance_sph_degree + 1);
    const real eps = R_(1e-9);
    /* per-hit gradients in `real`, SUMMED in double: the reference's float atomics add in a run-dependent order; the checker
     * does not reproduce one sample of that rounding noise (same rule as gut_oracle.c: orc_gut_render_bwd) */
    double* acc_d = (double*)calloc((size_t)N * 12 + 1, sizeof(double));
    double* acc_s = (double*)calloc((size_t)N * 3 * ncoef + 1, sizeof(double));
#pragma omp parallel
    {
        grt_hit* cands = (grt_hit*)malloc(sizeof(grt_hit) * (3 * (size_t)N + 3))   /* (trihexa: up to three offers per particle) */;
#pragma omp for schedule(dynamic, 8)
        for (uint32_t r = 0; r < nrays; ++r) {
            const v3 o = xform_point(ray_to_world12, v3_make(ray_o[3 * r], ray_o[3 * r + 1], ray_o[3 * r + 2]));
            const v3 d = xform_dir(ray_to_world12, v3_make(ray_d[3 * r], ray_d[3 * r + 1], ray_d[3 * r + 2]));
            grt_bwd_state b;
            b.T = 1; b.rad = v3_make(0, 0, 0); b.depth = 0;
            b.rad_fin = v3_make(rad[3 * r], rad[3 * r + 1], rad[3 * r + 2]);
            b.T_fin = 1 - dns[r];
            b.depth_fin = hit2[2 * r];
            const real maxHit = hit2[2 * r + 1];
            b.rad_grad = v3_make(g_rad[3 * r], g_rad[3 * r + 1], g_rad[3 * r + 2]);
            b.T_grad = -g_dns[r];
            b.depth_grad = g_hit[r];
            real tEnter, tExit;
            scene_interval(scene6, o, d, &tEnter, &tExit);
            real startT = r_max(0, tEnter - eps);
            const real endT = r_min(maxHit, tExit) + eps;
            const uint32_t n = ray_candidates(N, inst12, o, d, cands, r);
            grt_hit buf[GRT_MAX_K];
            uint32_t ndbg = 0;
            while (startT < endT) {
                const int k = trace_round(cands, n, startT + eps, endT, K, buf);
                if (k == 0) break;
                for (int i = 0; i < k; ++i) {
                    const uint32_t id = buf[i].id;
                    if (dbg_ids && ndbg < dbg_cap) dbg_ids[(size_t)r * dbg_cap + ndbg] = id;
                    ndbg++;
                    real gd[12] = {0}, gs[48] = {0};
                    process_hit_bwd(cfg, o, d, density12 + 12 * (size_t)id, sph + (size_t)id * 3 * ncoef, sph_deg, min_T, &b, gd, gs);
                    for (int c = 0; c < 11; ++c)
                        if (gd[c] != 0) {
#pragma omp atomic
                            acc_d[12 * (size_t)id + c] += (double)gd[c];
                        }
                    for (int c = 0; c < 3 * ncoef; ++c)
                        if (gs[c] != 0) {
#pragma omp atomic
                            acc_s[(size_t)id * 3 * ncoef + c] += (double)gs[c];
                        }
                    startT = r_max(startT, buf[i].t);
                }
            }
            if (dbg_count) dbg_count[r] = ndbg;
            if (round_shift) {
                /* Does this program process exactly the hits the FORWARD program processed (minus the last one and minus those whose
                 * box the ray enters after endT)?  Not necessarily: every hit that the endT clip removes lets its round take one
                 * more candidate, which moves every later round boundary, and a candidate whose box the ray had already left at the
                 * forward's boundary (tfar < tmin: never offered to the forward) can be offered here — or the other way round.
                 * Flagged rays are where a backward that REPLAYS the forward's hit list differs from one that traverses again. */
                uint32_t n_replay = 0, n_bwd = 0;
                uint64_t h_replay = 0, h_bwd = 0;   /* order-independent set signatures */
                real tl = r_max(0, tEnter - eps);
                int done = 0;
                while (!done && tl <= tExit) {
                    const int k = trace_round(cands, n, tl + eps, tExit + eps, K, buf);
                    if (k == 0) break;
                    for (int i = 0; i < k && !done; ++i) {
                        if (buf[i].t < endT && buf[i].tnear <= endT) { n_replay++; h_replay += (uint64_t)buf[i].id * 0x9E3779B97F4A7C15ull + 1; }
                        tl = r_max(tl, buf[i].t);
                        if (buf[i].t >= maxHit) done = 1;   /* the forward stopped at the hit it reported as its last */
                    }
                }
                real st = r_max(0, tEnter - eps);
                while (st < endT) {
                    const int k = trace_round(cands, n, st + eps, endT, K, buf);
                    if (k == 0) break;
                    for (int i = 0; i < k; ++i) { n_bwd++; h_bwd += (uint64_t)buf[i].id * 0x9E3779B97F4A7C15ull + 1; st = r_max(st, buf[i].t); }
                }
                round_shift[r] = (uint8_t)((n_replay != n_bwd) || (h_replay != h_bwd));
            }
        }
        free(cands);
    }
    for (size_t k = 0; k < (size_t)N * 12; ++k) g_density12[k] += (real)acc_d[k];
    for (size_t k = 0; k < (size_t)N * 3 * ncoef; ++k) g_sph[k] += (real)acc_s[k];
    free(acc_d); free(acc_s);
    return 0;
}

/* --------------------------------------------------------------------------------------------------------------------------------
 * The Slang pipelines with NEURAL HARMONIC FEATURES (render.pipeline_type referenceSlang / referenceSlangBwd, model.feature_type nht):
 * referenceSlangOptix.cu:103-186 — the same k = 16 rounds as the reference pipeline; per processed hit
 * particleDensityProcessHitFwdFromBuffer (gaussianParticles.slang:284-316, 404-425: hit(), integrateHit<false>, canonical intersection)
 * and particleFeaturesIntegrateFwdFromBuffer (neuralHarmonicFeaturesParticle.slang:213-228, 253-270) — and
 * referenceSlangBwdOptix.cu:70-185 — the backward program's rounds (as referenceBwdOptix.cu), per returned hit particleDensityHit,
 * particleFeaturesFromBuffer, particleFeaturesIntegrateBwdToBuffer (:272-320, lerp form un-blended front to back) and
 * particleDensityProcessHitBwdToBuffer (gaussianParticles.slang:420-479) with the canonical intersection's gradient.  The backward
 * functions are Slang autodiff output: this is the reverse mode of the restated forward, the same per-hit formulas that the 3DGUT
 * restatement checks against float64 torch.autograd (tests/golden/autograd_gut_nht.npz).
 * out_feat [nrays, ray_dim]; features [N, K]; g_features [N, K] accumulated into.
 * ------------------------------------------------------------------------------------------------------------------------------ */
int orc_grt_trace_nht_fwd(const GrtConfig* cfg, const int* nht, uint32_t N, const real* density12, const real* features, real min_T,
                          const real* inst12, const real* scene6, const real* ray_to_world12, uint32_t nrays, const real* ray_o,
                          const real* ray_d, real* out_feat, real* out_dns, real* out_hit2, real* out_cnt, int32_t* visibility,
                          uint32_t* dbg_ids, uint32_t* dbg_count, uint32_t dbg_cap) {
    orc_grt_set_primitive(cfg->primitive_type);
    g_custom_abs = cfg->primitive_type == 5;
    const int K = cfg->max_hits_per_trace > 0 ? cfg->max_hits_per_trace : 16;
    const int nr = orc_nht_ray_dim(nht), KF = nht[0];
    if (K > GRT_MAX_K || nr > ORC_NHT_MAX_DIM || nht[1] > ORC_NHT_MAX_DIM) return -1;
    const real eps = R_(1e-9);
    const orc_nht_tet tet = orc_nht_tetra();
#pragma omp parallel
    {
        grt_hit* cands = (grt_hit*)malloc(sizeof(grt_hit) * (3 * (size_t)N + 3))   /* (trihexa: up to three offers per particle) */;
#pragma omp for schedule(dynamic, 8)
        for (uint32_t r = 0; r < nrays; ++r) {
            const v3 o = xform_point(ray_to_world12, v3_make(ray_o[3 * r], ray_o[3 * r + 1], ray_o[3 * r + 2]));
            const v3 d = xform_dir(ray_to_world12, v3_make(ray_d[3 * r], ray_d[3 * r + 1], ray_d[3 * r + 2]));
            real T = 1, depth = 0, cnt = 0, acc[ORC_NHT_MAX_DIM];
            for (int i = 0; i < nr; ++i) acc[i] = 0;
            real tEnter, tExit;
            scene_interval(scene6, o, d, &tEnter, &tExit);
            real tLast = r_max(0, tEnter - eps);
            const uint32_t n = ray_candidates(N, inst12, o, d, cands, r);
            uint32_t ndbg = 0;
            grt_hit buf[GRT_MAX_K];
            while ((tLast <= tExit) && (T > min_T)) {
                const int k = trace_round(cands, n, tLast + eps, tExit + eps, K, buf);
                if (k == 0) break;
                for (int i = 0; i < k; ++i) {
                    if (T > min_T) {
                        const uint32_t id = buf[i].id;
                        const grt_particle p = load_particle(density12 + 12 * (size_t)id);
                        const v3 giscl = v3_make(1 / p.scl.x, 1 / p.scl.y, 1 / p.scl.z);
                        const v3 gro = v3_mul(giscl, v3_mul_rows(v3_sub(o, p.pos), &p.rotT));
                        const v3 grdu = v3_mul(giscl, v3_mul_rows(d, &p.rotT));
                        const v3 grd = v3_scale(grdu, 1 / r_sqrt(v3_dot(grdu, grdu)));
                        const v3 gcrod = v3_cross(grd, gro);
                        const real gres = particle_response(cfg->particle_kernel_degree, v3_dot(gcrod, gcrod));
                        const real alpha = r_min((real)cfg->particle_kernel_max_alpha, gres * p.density);
                        real w = 0;
                        if ((gres > (real)cfg->particle_kernel_min_response) && (alpha > (real)cfg->particle_kernel_min_alpha)) {
                            const v3 cg = v3_scale(grd, v3_dot(grd, v3_scale(gro, -1)));
                            const v3 P = v3_add(gro, cg);
                            const v3 grds = v3_mul(p.scl, cg);
                            w = alpha * T;
                            depth += r_sqrt(v3_dot(grds, grds)) * w;
                            T *= (1 - alpha);
                            if (w > 0) {
                                real wq[4], base[ORC_NHT_MAX_DIM], f[ORC_NHT_MAX_DIM];
                                orc_nht_weights(nht, &tet, P, wq);
                                orc_nht_features(nht, features + (size_t)KF * id, wq, base, f);
                                for (int c = 0; c < nr; ++c) acc[c] += f[c] * w;
#pragma omp atomic write
                                visibility[id] = 1;
                                cnt += 1;
                            }
                        }
                        tLast = r_max(tLast, buf[i].t);
                        if (dbg_ids && ndbg < dbg_cap) dbg_ids[(size_t)r * dbg_cap + ndbg] = id;
                        ndbg++;
                    }
                }
            }
            for (int c = 0; c < nr; ++c) out_feat[(size_t)nr * r + c] = acc[c];
            out_dns[r] = 1 - T;
            out_hit2[2 * r] = depth; out_hit2[2 * r + 1] = tLast;
            if (cfg->enable_hitcounts) out_cnt[r] = cnt;
            if (dbg_count) dbg_count[r] = ndbg;
        }
        free(cands);
    }
    return 0;
}

int orc_grt_trace_nht_bwd(const GrtConfig* cfg, const int* nht, uint32_t N, const real* density12, const real* features, real min_T,
                          const real* inst12, const real* scene6, const real* ray_to_world12, uint32_t nrays, const real* ray_o,
                          const real* ray_d, const real* feat, const real* dns, const real* hit2, const real* g_feat, const real* g_dns,
                          const real* g_hit, real* g_density12, real* g_features) {
    orc_grt_set_primitive(cfg->primitive_type);
    g_custom_abs = cfg->primitive_type == 5;
    (void)min_T;
    const int K = cfg->max_hits_per_trace > 0 ? cfg->max_hits_per_trace : 16;
    const int nr = orc_nht_ray_dim(nht), KF = nht[0];
    if (K > GRT_MAX_K || nr > ORC_NHT_MAX_DIM || nht[1] > ORC_NHT_MAX_DIM) return -1;
    const real eps = R_(1e-9);
    const orc_nht_tet tet = orc_nht_tetra();
    double* acc_d = (double*)calloc((size_t)N * 12 + 1, sizeof(double));
    double* acc_f = (double*)calloc((size_t)N * KF + 1, sizeof(double));
#pragma omp parallel
    {
        grt_hit* cands = (grt_hit*)malloc(sizeof(grt_hit) * (3 * (size_t)N + 3))   /* (trihexa: up to three offers per particle) */;
#pragma omp for schedule(dynamic, 8)
        for (uint32_t r = 0; r < nrays; ++r) {
            const v3 o = xform_point(ray_to_world12, v3_make(ray_o[3 * r], ray_o[3 * r + 1], ray_o[3 * r + 2]));
            const v3 d = xform_dir(ray_to_world12, v3_make(ray_d[3 * r], ray_d[3 * r + 1], ray_d[3 * r + 2]));
            real Cb[ORC_NHT_MAX_DIM], gC[ORC_NHT_MAX_DIM];
            for (int c = 0; c < nr; ++c) { Cb[c] = feat[(size_t)nr * r + c]; gC[c] = g_feat[(size_t)nr * r + c]; }
            real Tb = 1 - dns[r], gT = -g_dns[r], Db = hit2[2 * r], gD = g_hit ? g_hit[r] : 0;
            const real maxHit = hit2[2 * r + 1];
            real tEnter, tExit;
            scene_interval(scene6, o, d, &tEnter, &tExit);
            real startT = r_max(0, tEnter - eps);
            const real endT = r_min(maxHit, tExit) + eps;
            const uint32_t n = ray_candidates(N, inst12, o, d, cands, r);
            grt_hit buf[GRT_MAX_K];
            while (startT < endT) {
                const int k = trace_round(cands, n, startT + eps, endT, K, buf);
                if (k == 0) break;
                for (int i = 0; i < k; ++i) {
                    const uint32_t id = buf[i].id;
                    startT = r_max(startT, buf[i].t);
                    const grt_particle p = load_particle(density12 + 12 * (size_t)id);
                    const v3 gscl = p.scl;
                    const v3 giscl = v3_make(1 / gscl.x, 1 / gscl.y, 1 / gscl.z);
                    const v3 gposc = v3_sub(o, p.pos);
                    const v3 gposcr = v3_mul_rows(gposc, &p.rotT);
                    const v3 gro = v3_mul(giscl, gposcr);
                    const v3 rdr = v3_mul_rows(d, &p.rotT);
                    const v3 grdu = v3_mul(giscl, rdr);
                    const v3 grd = v3_scale(grdu, 1 / r_sqrt(v3_dot(grdu, grdu)));
                    const v3 gcrod = v3_cross(grd, gro);
                    const real gray = v3_dot(gcrod, gcrod);
                    const real gres = particle_response(cfg->particle_kernel_degree, gray);
                    const real alpha = r_min((real)cfg->particle_kernel_max_alpha, gres * p.density);
                    if (!((gres > (real)cfg->particle_kernel_min_response) && (alpha > (real)cfg->particle_kernel_min_alpha))) continue;
                    const real pdot = v3_dot(grd, v3_scale(gro, -1));
                    const v3 grdd = v3_scale(grd, pdot);
                    const v3 P = v3_add(gro, grdd);
                    const v3 grds = v3_mul(gscl, grdd);
                    const real gsq = v3_dot(grds, grds);
                    const real hitT = r_sqrt(gsq);
                    const real* row = features + (size_t)KF * id;
                    real wq[4], base[ORC_NHT_MAX_DIM], f[ORC_NHT_MAX_DIM], gf[ORC_NHT_MAX_DIM], g_row[ORC_NHT_MAX_DIM * 4];
                    orc_nht_weights(nht, &tet, P, wq);
                    orc_nht_features(nht, row, wq, base, f);
                    const real w = 1 / (1 - alpha);
                    real dalpha = 0;
                    if (alpha > 0) {
                        for (int c = 0; c < nr; ++c) {
                            Cb[c] = (Cb[c] - f[c] * alpha) * w;
                            dalpha += (f[c] - Cb[c]) * gC[c];
                            gf[c] = alpha * gC[c];
                            gC[c] *= (1 - alpha);
                        }
                    } else {
                        for (int c = 0; c < nr; ++c) gf[c] = 0;
                    }
                    const v3 dP = orc_nht_features_bwd(nht, &tet, row, wq, base, gf, g_row);
                    for (int c = 0; c < KF; ++c)
                        if (g_row[c] != 0) {
#pragma omp atomic
                            acc_f[(size_t)KF * id + c] += (double)g_row[c];
                        }
                    Tb *= w;
                    Db = (Db - hitT * alpha) * w;
                    dalpha += (hitT - Db) * gD - Tb * gT;
                    const real ddepth = alpha * gD;
                    gD *= (1 - alpha);
                    gT *= (1 - alpha);
                    real gd[12] = {0};
                    real dres = 0, ddens = 0;
                    if (gres * p.density < (real)cfg->particle_kernel_max_alpha) { dres = p.density * dalpha; ddens = gres * dalpha; }
                    gd[3] = ddens;
                    const real grayGrd = particle_response_grd(cfg->particle_kernel_degree, gray, gres, dres);
                    const v3 grdsGrd = gsq > 0 ? v3_scale(grds, ddepth / hitT) : v3_make(0, 0, 0);
                    const v3 gsclHit = v3_mul(grdd, grdsGrd);
                    const real sdot = v3_dot(v3_mul(grdsGrd, gscl), grd);
                    const real gdP = v3_dot(grd, dP);
                    const v3 grdHit = v3_add(v3_sub(v3_scale(v3_mul(gscl, grdsGrd), pdot), v3_scale(gro, sdot)), v3_sub(v3_scale(dP, pdot), v3_scale(gro, gdP)));
                    const v3 groHit = v3_add(v3_scale(grd, -sdot), v3_sub(dP, v3_scale(grd, gdP)));
                    const v3 gcrodGrd = v3_scale(gcrod, 2 * grayGrd);
                    const v3 grdGrd = v3_make(gcrodGrd.z * gro.y - gcrodGrd.y * gro.z, gcrodGrd.x * gro.z - gcrodGrd.z * gro.x, gcrodGrd.y * gro.x - gcrodGrd.x * gro.y);
                    const v3 groGrd = v3_make(gcrodGrd.y * grd.z - gcrodGrd.z * grd.y, gcrodGrd.z * grd.x - gcrodGrd.x * grd.z, gcrodGrd.x * grd.y - gcrodGrd.y * grd.x);
                    const v3 groTot = v3_add(groGrd, groHit);
                    const v3 gsclGro = v3_mul(v3_make(-gposcr.x / (gscl.x * gscl.x), -gposcr.y / (gscl.y * gscl.y), -gposcr.z / (gscl.z * gscl.z)), groTot);
                    const v3 gposcrGrd = v3_mul(giscl, groTot);
                    const v3 gposcGrd = matmul_bw_vec(&p.rotT, gposcrGrd);
                    const v4 gq1 = matmul_bw_quat(gposc, gposcrGrd, p.quat);
                    gd[0] = -gposcGrd.x; gd[1] = -gposcGrd.y; gd[2] = -gposcGrd.z;
                    const v3 grduGrd = v3_safe_normalize_bw(grdu, v3_add(grdGrd, grdHit));
                    const v3 sclGrd = v3_add(v3_add(gsclHit, gsclGro),
                                             v3_mul(v3_make(-rdr.x / (gscl.x * gscl.x), -rdr.y / (gscl.y * gscl.y), -rdr.z / (gscl.z * gscl.z)), grduGrd));
                    gd[8] = sclGrd.x; gd[9] = sclGrd.y; gd[10] = sclGrd.z;
                    const v4 gq2 = matmul_bw_quat(d, v3_mul(giscl, grduGrd), p.quat);
                    gd[4] = gq1.x + gq2.x; gd[5] = gq1.y + gq2.y; gd[6] = gq1.z + gq2.z; gd[7] = gq1.w + gq2.w;
                    for (int c = 0; c < 11; ++c)
                        if (gd[c] != 0) {
#pragma omp atomic
                            acc_d[12 * (size_t)id + c] += (double)gd[c];
                        }
                }
            }
        }
        free(cands);
    }
    for (size_t k = 0; k < (size_t)N * 12; ++k) g_density12[k] += (real)acc_d[k];
    for (size_t k = 0; k < (size_t)N * KF; ++k) g_features[k] += (real)acc_f[k];
    free(acc_d); free(acc_f);
    return 0;
}

/* Debug / analysis aid: the full candidate list of ONE ray, sorted by (t, id) — what every round of the forward and the
 * backward selects from.  ray in ray space; returns the number of candidates written (<= cap). */
int orc_grt_ray_candidates(uint32_t N, const real* inst12, const real* ray_to_world12, const real* ray_o3, const real* ray_d3, uint32_t cap,
                           uint32_t* out_id, real* out_t, real* out_tnear, real* out_tfar) {
    const v3 o = xform_point(ray_to_world12, v3_make(ray_o3[0], ray_o3[1], ray_o3[2]));
    const v3 d = xform_dir(ray_to_world12, v3_make(ray_d3[0], ray_d3[1], ray_d3[2]));
    grt_hit* cands = (grt_hit*)malloc(sizeof(grt_hit) * (3 * (size_t)N + 3))   /* (trihexa: up to three offers per particle) */;
    const uint32_t n = ray_candidates(N, inst12, o, d, cands, 0xFFFFFFFFu /* no prefilter: a ray outside the frame's packets */);
    uint32_t k = 0;
    for (; k < n && k < cap; ++k) { out_id[k] = cands[k].id; out_t[k] = cands[k].t; out_tnear[k] = cands[k].tnear; out_tfar[k] = cands[k].tfar; }
    free(cands);
    return (int)k;
}

/* =====================================================================================================================
 * Hybrid mesh + Gaussian path tracing (BASELINE config 5): threedgrut_playground/src/kernels/cuda/
 * playgroundKernel.cu:39-157 (__raygen__rg path loop), :159-251 (refract / handleMirror / handleGlass / handlePBR / handleDiffuse),
 * :253-352 (normals, __closesthit__ch, __miss__ms); include/playground/kernels/cuda/trace.cuh:175-259 (traceMesh, traceGaussians,
 * getBackgroundColor), materials.cuh:34-442 (get_diffuse_color, tangent frames, GGX importance sampling, alpha test, Cook-Torrance
 * sampling), rng.cuh:21-78 (tea, lcg, rnd, rnd_pcg3d); threedgrt_tracer/include/3dgrt/kernels/cuda/3dgrtTracer.cuh:137-204
 * (traceVolumetricGS: the k = 16 rounds of the forward program on a sub-interval of the ray, continuing the ray's transmittance).
 * PINNED by tests/golden/playground.npz: the reference's own programs compiled on the host over an emulated OptiX
 * (oracle/ref/ref_playground.cpp).  What the emulation defines (not the reference): the triangle intersection of OptiX's built-in
 * triangle GAS — Moeller-Trumbore over every triangle, ties to the lower triangle index — and the texture filter: bilinear with float
 * weights, clamp-to-edge, normalised coordinates (CUDA hardware filters with 8 fractional bits; cutexture.h:54-60 sets the modes).
 * ===================================================================================================================== */
typedef struct {
    const float* data;       /* [height, width, channels] or NULL (no texture) */
    int32_t height, width, channels;
} orc_texture;
typedef struct {             /* PBRMaterial, pipelineParameters.h:26-52 */
    orc_texture diffuse, emissive, metallic_roughness, normal;   /* 4, 4, 2, 4 channels */
    float diffuse_factor[4], emissive_factor[3];
    float metallic_factor, roughness_factor, transmission_factor, ior, alpha_cutoff;
    uint32_t alpha_mode;     /* GltfAlphaMode: 0 opaque, 1 blend, 2 mask */
} orc_material;
typedef struct {
    uint32_t num_vertices, num_faces;
    const float* vertices;            /* [V,3] */
    const int32_t* triangles;         /* [F,3] */
    const float* vertex_normals;      /* [V,3] */
    const float* vertex_tangents;     /* [V,3] */
    const uint8_t* vertex_has_tangents; /* [V] */
    const int32_t* prim_type;         /* [F] PlaygroundPrimitiveTypes */
    const float* mat_uv;              /* [F,3,2] */
    const int32_t* mat_id;            /* [F] */
    const float* refractive_index;    /* [F] */
    uint32_t num_materials;
    const orc_material* materials;
    orc_texture envmap;               /* 4 channels, or data == NULL: black */
    float envmap_offset[2];
} orc_mesh;

static int tri_intersect(const orc_mesh* m, uint32_t f, v3 o, v3 d, real tmin, real tmax, real* t_out, real* u_out, real* v_out) {
    const int32_t* tr = m->triangles + 3 * (size_t)f;
    const float* p0 = m->vertices + 3 * (size_t)tr[0]; const float* p1 = m->vertices + 3 * (size_t)tr[1]; const float* p2 = m->vertices + 3 * (size_t)tr[2];
    const v3 v0 = v3_make(p0[0], p0[1], p0[2]);
    const v3 e1 = v3_sub(v3_make(p1[0], p1[1], p1[2]), v0), e2 = v3_sub(v3_make(p2[0], p2[1], p2[2]), v0);
    const v3 pv = v3_cross(d, e2);
    const real det = v3_dot(e1, pv);
    if (!(r_fabs(det) > R_(1e-20))) return 0;
    const real inv = 1 / det;
    const v3 tv = v3_sub(o, v0);
    const real u = v3_dot(tv, pv) * inv;
    if (u < 0 || u > 1) return 0;
    const v3 q = v3_cross(tv, e1);
    const real v = v3_dot(d, q) * inv;
    if (v < 0 || u + v > 1) return 0;
    const real t = v3_dot(e2, q) * inv;
    if (!(t > tmin && t < tmax)) return 0;
    *t_out = t; *u_out = u; *v_out = v;
    return 1;
}

/* tex2D with the modes of cutexture.h:54-60: normalised coordinates, clamp-to-edge, bilinear, element type float */
static void tex_fetch(const orc_texture* t, real u, real v, real out[4]) {
    out[0] = out[1] = out[2] = out[3] = 0;
    if (!t->data) return;
    const real x = u * (real)t->width - R_(0.5), y = v * (real)t->height - R_(0.5);
    const real fx = r_floor(x), fy = r_floor(y);
    const real ax = x - fx, ay = y - fy;
    int x0 = (int)fx, x1 = (int)fx + 1, y0 = (int)fy, y1 = (int)fy + 1;
    x0 = x0 < 0 ? 0 : (x0 >= t->width ? t->width - 1 : x0); x1 = x1 < 0 ? 0 : (x1 >= t->width ? t->width - 1 : x1);
    y0 = y0 < 0 ? 0 : (y0 >= t->height ? t->height - 1 : y0); y1 = y1 < 0 ? 0 : (y1 >= t->height ? t->height - 1 : y1);
    for (int c = 0; c < t->channels && c < 4; ++c) {
        const real t00 = t->data[((size_t)y0 * t->width + x0) * t->channels + c], t10 = t->data[((size_t)y0 * t->width + x1) * t->channels + c];
        const real t01 = t->data[((size_t)y1 * t->width + x0) * t->channels + c], t11 = t->data[((size_t)y1 * t->width + x1) * t->channels + c];
        out[c] = (1 - ay) * ((1 - ax) * t00 + ax * t10) + ay * ((1 - ax) * t01 + ax * t11);
    }
}

/* traceVolumetricGS on [tmin, tmax] of the ray (o, d), continuing the ray's transmittance T and radiance */
static void trace_segment(const GrtConfig* cfg, uint32_t N, const real* density12, const real* sph, int sph_deg, real min_T, const real* inst12,
                          const real* scene6, v3 o, v3 d, real tmin, real tmax, grt_hit* cands, real* T, v3* rad) {
    const int K = cfg->max_hits_per_trace > 0 ? cfg->max_hits_per_trace : 16;
    const int ncoef = (cfg->particle_radiance_sph_degree + 1) * (cfg->particle_radiance_sph_degree + 1);
    const real eps = R_(1e-9);
    real t0, t1;
    scene_interval(scene6, o, d, &t0, &t1);
    t0 = r_max(t0, tmin); t1 = r_min(t1, tmax);
    real tLast = r_max(0, t0 - eps);
    grt_ray_state s; s.T = *T; s.rad = *rad; s.depth = 0; s.normal = v3_make(0, 0, 0);
    const uint32_t n = ray_candidates(N, inst12, o, d, cands, 0xFFFFFFFFu /* no prefilter: a ray outside the frame's packets */);
    grt_hit buf[GRT_MAX_K];
    while ((tLast <= t1) && (s.T > min_T)) {
        const int k = trace_round(cands, n, tLast + eps, t1 + eps, K, buf);
        if (k == 0) break;
        for (int i = 0; i < k; ++i) {
            if (s.T > min_T) {
                process_hit(cfg, o, d, density12 + 12 * (size_t)buf[i].id, sph + (size_t)buf[i].id * 3 * ncoef, sph_deg, &s, 0);
                tLast = r_max(tLast, buf[i].t);
            }
        }
    }
    *T = s.T; *rad = s.rad;
}

static int refract_dir(v3* out_dir, v3 ray_d, v3 normal, real etai_over_etat) {   /* playgroundKernel.cu:159-188 */
    real ri;
    if (v3_dot(ray_d, normal) < 0) ri = 1 / etai_over_etat;
    else { ri = etai_over_etat; normal = v3_scale(normal, -1); }
    const real cos_theta = r_min(v3_dot(v3_scale(ray_d, -1), normal), 1);
    const real sin_theta = r_sqrt(1 - cos_theta * cos_theta);
    if (!(ri * sin_theta <= 1)) return 0;
    const v3 perp = v3_scale(v3_add(ray_d, v3_scale(normal, cos_theta)), ri);
    const v3 par = v3_scale(normal, -r_sqrt(r_fabs(1 - v3_dot(perp, perp))));
    *out_dir = v3_safe_normalize(v3_add(perp, par));
    return 1;
}
static v3 mirror_dir(v3 ray_d, v3 normal) {   /* :190-198: -reflect(d, n') with reflect(x, n) = 2 n (n.x) - x (mathUtils.cuh:384-387) */
    const v3 n = v3_dot(ray_d, normal) < 0 ? normal : v3_scale(normal, -1);
    return v3_safe_normalize(v3_sub(ray_d, v3_scale(n, 2 * v3_dot(n, ray_d))));
}

/* ---- rng.cuh ---- */
static uint32_t pg_tea16(uint32_t val0, uint32_t val1) {   /* tea<16>, :21-35 */
    uint32_t v0 = val0, v1 = val1, s0 = 0;
    for (int n = 0; n < 16; ++n) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}
static real pg_rnd(uint32_t* prev) {   /* lcg + rnd, :38-54 */
    *prev = 1664525u * *prev + 1013904223u;
    return (real)((float)(*prev & 0x00FFFFFFu) / (float)0x01000000);
}
static v3 pg_rnd_pcg3d(uint32_t x, uint32_t y, uint32_t z) {   /* :62-76 */
    x = x * 1664525u + 1013904223u; y = y * 1664525u + 1013904223u; z = z * 1664525u + 1013904223u;
    x += y * z; y += z * x; z += x * y;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16;
    x += y * z; y += z * x; z += x * y;
    /* make_float3(v.x, v.y, v.z) * (1.0 / float(0xffffffffu)): uint -> float conversions, then a float scale by 2^-32 (float(0xffffffff) = 2^32) */
    const float k = (float)(1.0 / (double)(float)0xffffffffu);
    return v3_make((real)((float)x * k), (real)((float)y * k), (real)((float)z * k));
}

/* ---- materials.cuh ---- */
/* (the reference's trigonometry is single precision: asin / acos / sin / cos of float arguments) */
static real pg_sin(real x) { return sizeof(real) == 4 ? (real)sinf((float)x) : (real)sin((double)x); }
static real pg_cos(real x) { return sizeof(real) == 4 ? (real)cosf((float)x) : (real)cos((double)x); }
static real pg_asin(real x) { return sizeof(real) == 4 ? (real)asinf((float)x) : (real)asin((double)x); }
static real pg_acos(real x) { return sizeof(real) == 4 ? (real)acosf((float)x) : (real)acos((double)x); }
#define PG_PBR_EPS R_(1e-6)
#define PG_PI R_(3.141592654)
static v3 pg_normalize(v3 v) {   /* :79-82 (not safe_normalize: threshold 1e-6 on the squared norm) */
    const real n = v3_dot(v, v);
    return (n > PG_PBR_EPS) ? v3_scale(v, 1 / r_sqrt(n)) : v;
}
static real pg_clamp01(real x) { return r_min(1, r_max(0, x)); }
static real pg_pdot(v3 a, v3 b) { return pg_clamp01(v3_dot(a, b)); }   /* positive_dot, :68-71 */

typedef struct {   /* what the closest-hit program knows about the hit (OptiX getters) */
    uint32_t tri;
    real bu, bv;       /* optixGetTriangleBarycentrics */
} pg_hit;

static void pg_tex_coords(const orc_mesh* m, const pg_hit* h, real* u, real* v) {   /* :41-48 */
    const float* uv = m->mat_uv + 6 * (size_t)h->tri;
    const real w0 = 1 - h->bu - h->bv;
    *u = w0 * uv[0] + h->bu * uv[2] + h->bv * uv[4];
    *v = w0 * uv[1] + h->bu * uv[3] + h->bv * uv[5];
}
static v3 pg_diffuse_color(const orc_mesh* m, const pg_hit* h, uint32_t opts, v3 ray_d, v3 normal) {   /* get_diffuse_color, :34-66 */
    const orc_material* mat = &m->materials[m->mat_id[h->tri]];
    real u, v;
    pg_tex_coords(m, h, &u, &v);
    v3 diffuse = v3_make(mat->diffuse_factor[0], mat->diffuse_factor[1], mat->diffuse_factor[2]);
    if (mat->diffuse.data && !(opts & 4u)) {
        real t[4];
        tex_fetch(&mat->diffuse, u, v, t);
        diffuse = v3_mul(v3_make(t[0], t[1], t[2]), diffuse);
    }
    return v3_scale(diffuse, r_fabs(v3_dot(ray_d, normal)));
}
static v3 pg_normal_space(const orc_mesh* m, const pg_hit* h, v3 normal, v3 dir) {   /* compute_normal_space, :84-135 */
    const int32_t* tr = m->triangles + 3 * (size_t)h->tri;
    v3 tangent;
    if (m->vertex_has_tangents[tr[0]] && m->vertex_has_tangents[tr[1]] && m->vertex_has_tangents[tr[2]]) {   /* get_smooth_tangent */
        const float* t0 = m->vertex_tangents + 3 * (size_t)tr[0]; const float* t1 = m->vertex_tangents + 3 * (size_t)tr[1];
        const float* t2 = m->vertex_tangents + 3 * (size_t)tr[2];
        const real w0 = 1 - h->bu - h->bv;
        tangent = v3_make(w0 * t0[0] + h->bu * t1[0] + h->bv * t2[0], w0 * t0[1] + h->bu * t1[1] + h->bv * t2[1], w0 * t0[2] + h->bu * t1[2] + h->bv * t2[2]);
        { const real len = r_sqrt(v3_dot(tangent, tangent)); tangent = v3_make(tangent.x / len, tangent.y / len, tangent.z / len); }   /* `/= length(...)` */
    } else if (r_fabs(normal.x) > r_fabs(normal.z)) tangent = v3_make(-normal.y, normal.x, 0);
    else tangent = v3_make(0, -normal.z, normal.y);
    tangent = pg_normalize(tangent);
    const v3 bitangent = pg_normalize(v3_cross(normal, tangent));
    return pg_normalize(v3_make(tangent.x * dir.x + bitangent.x * dir.y + normal.x * dir.z, tangent.y * dir.x + bitangent.y * dir.y + normal.y * dir.z,
                                tangent.z * dir.x + bitangent.z * dir.y + normal.z * dir.z));
}
static v3 pg_sample_diffuse(const orc_mesh* m, const pg_hit* h, v3 normal, real theta_seed, real phi_seed) {   /* :137-149 */
    const real theta = pg_asin(theta_seed), phi = R_(2.0) * PG_PI * phi_seed;
    return pg_normal_space(m, h, normal, v3_make(pg_sin(theta) * pg_cos(phi), pg_sin(theta) * pg_sin(phi), pg_cos(theta)));
}
static v3 pg_sample_specular(const orc_mesh* m, const pg_hit* h, v3 normal, real theta_seed, real phi_seed, real roughness) {   /* :151-164 */
    const real alpha = roughness * roughness;
    const real theta = pg_acos(r_sqrt((R_(1.0) - theta_seed) / (R_(1.0) + (alpha * alpha - R_(1.0)) * theta_seed)));
    const real phi = R_(2.0) * PG_PI * phi_seed;
    return pg_normal_space(m, h, normal, v3_make(pg_sin(theta) * pg_cos(phi), pg_sin(theta) * pg_sin(phi), pg_cos(theta)));
}
static real pg_ggx_d(v3 hv, v3 normal, real roughness) {   /* trowbridge_reitz_ggx, :184-193 */
    const real alpha = roughness * roughness, a2 = alpha * alpha, ndh = pg_pdot(normal, hv);
    const real denom = ndh * ndh * (a2 - 1) + 1;
    return a2 / r_max(PG_PI * denom * denom, PG_PBR_EPS);
}
static real pg_g1(real ndv, real roughness) {   /* geometry_schlick_ggx, :196-201 */
    const real alpha = R_(0.5) * roughness * roughness;
    return ndv / r_max(ndv * (1 - alpha) + alpha, PG_PBR_EPS);
}
static v3 pg_fresnel(real cosine, v3 f0) {   /* fresnel_schlick, :212-215 */
    const real p = r_pow(1 - cosine, 5);
    return v3_make(f0.x + (1 - f0.x) * p, f0.y + (1 - f0.y) * p, f0.z + (1 - f0.z) * p);
}
static v3 pg_lerp3(v3 a, v3 b, real t) { return v3_make(a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z)); }

typedef struct {   /* the fields of HybridRayPayload the PBR branch writes */
    v3 bsdf, emissive;
    uint32_t pbr_bounces, rnd_seed;
} pg_pbr_state;

/* sampled_microfacet_brdf, :231-340 */
static v3 pg_microfacet(const orc_mesh* m, const pg_hit* h, v3 wo, v3 normal, v3 base, real metalness, real roughness, real transmission, real ior,
                        uint32_t px, uint32_t py, uint32_t frame, pg_pbr_state* st, v3* next_dir) {
    const real fr = R_(0.5);
    const v3 rnd = pg_rnd_pcg3d(px, py, frame + st->pbr_bounces);
    const real phi_seed = rnd.x, theta_seed = rnd.y, ray_prob = rnd.z;
    v3 out, L;
    v3 f0 = v3_make(R_(0.16) * fr * fr, R_(0.16) * fr * fr, R_(0.16) * fr * fr);
    f0 = pg_lerp3(f0, base, metalness);
    if ((ray_prob < R_(0.5)) && ((R_(2.0) * ray_prob) < transmission)) {   /* transmissive */
        const real front = v3_dot(wo, normal);
        const v3 fn = front >= 0 ? normal : v3_scale(normal, -1);
        const real eta = front >= 0 ? 1 / ior : ior;
        const v3 H = pg_sample_specular(m, h, fn, theta_seed, phi_seed, roughness);
        {   /* pbr_refract(-wo, H, eta), :217-222 */
            const v3 wi = v3_scale(wo, -1);
            const real ndi = v3_dot(H, wi);
            const real k = 1 - eta * eta * (1 - ndi * ndi);
            L = (k < 0) ? v3_make(0, 0, 0) : v3_sub(v3_scale(wi, eta), v3_scale(H, eta * ndi + r_sqrt(k)));
        }
        const real ndo = pg_pdot(fn, wo), ndl = pg_pdot(v3_scale(fn, -1), L), ndh = pg_pdot(fn, H), odh = pg_pdot(wo, H);
        const v3 Fv = pg_fresnel(odh, f0);
        const real G = pg_g1(ndo, roughness) * pg_g1(ndl, roughness);
        (void)pg_ggx_d(H, fn, roughness);
        const real k = G * odh / r_max(ndh * ndo, R_(0.001));
        out = v3_make(base.x * (1 - Fv.x) * k, base.y * (1 - Fv.y) * k, base.z * (1 - Fv.z) * k);
    } else if ((ray_prob < R_(0.5)) && ((R_(2.0) * ray_prob) >= transmission)) {   /* diffuse */
        L = pg_sample_diffuse(m, h, normal, theta_seed, phi_seed);
        const v3 H = pg_normalize(v3_add(wo, L));
        const v3 Fv = pg_fresnel(pg_pdot(wo, H), f0);
        const real nm = 1 - metalness;
        out = v3_make((1 - Fv.x) * nm * base.x, (1 - Fv.y) * nm * base.y, (1 - Fv.z) * nm * base.z);
    } else {   /* specular */
        const v3 H = pg_sample_specular(m, h, normal, theta_seed, phi_seed, roughness);
        const v3 mwo = v3_scale(wo, -1);
        L = v3_sub(mwo, v3_scale(H, R_(2.0) * v3_dot(H, mwo)));
        const real ndo = pg_pdot(normal, wo), ndh = pg_pdot(normal, H), ndl = pg_pdot(normal, L), odh = pg_pdot(wo, H);
        const v3 Fv = pg_fresnel(odh, f0);
        const real G = pg_g1(ndo, roughness) * pg_g1(ndl, roughness);
        const real k = G * odh / r_max(ndh * ndo, R_(0.001));
        out = v3_scale(Fv, k);
    }
    *next_dir = L;
    st->pbr_bounces += 1;
    return v3_scale(out, 2);   /* compensates for splitting diffuse and specular */
}

/* sampled_cook_torrance_brdf, :344-440: material fetch (factors x textures), normal map, alpha test, the sampled BRDF */
static void pg_cook_torrance(const orc_mesh* m, const pg_hit* h, uint32_t opts, v3 ray_d, v3 normal, uint32_t px, uint32_t py, uint32_t frame,
                             pg_pbr_state* st, v3* new_dir) {
    const v3 wo = pg_normalize(v3_scale(ray_d, -1));
    const orc_material* mat = &m->materials[m->mat_id[h->tri]];
    const int notex = (opts & 4u) != 0;
    real u, v, t[4];
    pg_tex_coords(m, h, &u, &v);
    v3 diffuse = v3_make(mat->diffuse_factor[0], mat->diffuse_factor[1], mat->diffuse_factor[2]);
    real alpha = mat->diffuse_factor[3];
    if (mat->diffuse.data && !notex) {
        tex_fetch(&mat->diffuse, u, v, t);
        diffuse = v3_mul(v3_make(t[0], t[1], t[2]), diffuse);
        alpha *= t[3];
    }
    v3 emissive = v3_make(mat->emissive_factor[0], mat->emissive_factor[1], mat->emissive_factor[2]);
    if (mat->emissive.data && !notex) {
        tex_fetch(&mat->emissive, u, v, t);
        emissive = v3_mul(v3_make(t[0], t[1], t[2]), emissive);
    }
    real metallic = mat->metallic_factor, roughness = mat->roughness_factor;
    if (mat->metallic_roughness.data && !notex) {
        tex_fetch(&mat->metallic_roughness, u, v, t);
        metallic = t[0] * mat->metallic_factor;
        roughness = t[1] * mat->roughness_factor;
    }
    if (mat->normal.data && !notex) {
        tex_fetch(&mat->normal, u, v, t);
        normal = pg_normal_space(m, h, normal, v3_make(t[0], t[1], t[2]));
    }
    /* alpha_test, :166-181 */
    int pass = 1;
    if (mat->alpha_mode == 1u) pass = alpha > pg_rnd(&st->rnd_seed);
    else if (mat->alpha_mode == 2u) pass = alpha > (real)mat->alpha_cutoff;
    if (!pass) { *new_dir = ray_d; return; }
    const v3 scatter = pg_microfacet(m, h, wo, normal, diffuse, metallic, roughness, mat->transmission_factor, mat->ior, px, py, frame, st, new_dir);
    st->bsdf = v3_make(r_max(scatter.x, 0), r_max(scatter.y, 0), r_max(scatter.z, 0));
    st->emissive = emissive;
}

static v3 pg_background(const orc_mesh* m, v3 d) {   /* getBackgroundColor, trace.cuh:233-257 */
    const real rotY = (real)m->envmap_offset[0] * R_(2.0) * R_(3.14159265358979323846), rotX = R_(2.0) * (real)m->envmap_offset[1] * R_(3.14159265358979323846);
    const real cy = pg_cos(rotY), sy = pg_sin(rotY), cx = pg_cos(rotX), sx = pg_sin(rotX);
    const v3 r1 = v3_make(d.x * cy - d.z * sy, d.y, d.x * sy + d.z * cy);
    const v3 r2 = v3_make(r1.x, r1.y * cx - r1.z * sx, r1.y * sx + r1.z * cx);
    const real theta = r_atan2(r2.x, r2.z);
    const real phi = R_(3.14159265358979323846) * R_(0.5) - pg_acos(r_min(1, r_max(-1, r2.y)));
    const real u = (theta + R_(3.14159265358979323846)) * (R_(0.5) * R_(0.318309886183790671538));
    const real v = R_(0.5) * (1 + pg_sin(phi));
    real t[4];
    tex_fetch(&m->envmap, u, v, t);
    return v3_make(t[0], t[1], t[2]);
}

/* rays [height*width,3] in ray space, pixel (x, y) = ray y*width + x (the random streams are seeded per pixel); to trace a SUBSET of a
 * larger launch pass pixel_xy [rays,2] (the launch coordinates of each ray) and launch_width (0: the frame is the launch);
 * ray_max_t [rays] or NULL (= 1e30); opts: PlaygroundRenderOptions (1 smooth normals, 2 Gaussian tracing off, 4 PBR textures off).
 * out_rgba [rays,4], out_last_ray [rays,6] (origin, direction of the last traced segment, world space), out_bounces [rays] (mirror bounces). */
int orc_grt_hybrid_trace(const GrtConfig* cfg, uint32_t N, const real* density12, const real* sph, int sph_deg, real min_T,
                         const real* inst12, const real* scene6, const real* ray_to_world12, uint32_t width, uint32_t height, const real* ray_o,
                         const real* ray_d, const real* ray_max_t, const orc_mesh* mesh, uint32_t opts, uint32_t max_pbr_bounces,
                         uint32_t frame_number, const uint32_t* pixel_xy, uint32_t launch_width, real* out_rgba, real* out_last_ray,
                         uint32_t* out_bounces) {
    orc_grt_set_primitive(cfg->primitive_type);
    const uint32_t nrays = width * height;
    const uint32_t seed_width = launch_width ? launch_width : width;
#pragma omp parallel
    {
        grt_hit* cands = (grt_hit*)malloc(sizeof(grt_hit) * (3 * (size_t)N + 3))   /* (trihexa: up to three offers per particle) */;
#pragma omp for schedule(dynamic, 8)
        for (uint32_t r = 0; r < nrays; ++r) {
            const uint32_t px = pixel_xy ? pixel_xy[2 * r] : r % width, py = pixel_xy ? pixel_xy[2 * r + 1] : r / width;
            v3 rayOri = xform_point(ray_to_world12, v3_make(ray_o[3 * r], ray_o[3 * r + 1], ray_o[3 * r + 2]));
            v3 rayDir = xform_dir(ray_to_world12, v3_make(ray_d[3 * r], ray_d[3 * r + 1], ray_d[3 * r + 2]));
            const real ray_t_max = ray_max_t ? ray_max_t[r] : R_(1e30);
            /* payload (playgroundKernel.cu:51-68) and the ray's running volumetric state (RayData) */
            v3 accC = v3_make(0, 0, 0), direct = v3_make(0, 0, 0), thr = v3_make(1, 1, 1);
            real accA = 0;
            uint32_t numBounces = 0, timeout = 0;
            pg_pbr_state pbr; pbr.pbr_bounces = 0; pbr.rnd_seed = pg_tea16(seed_width * py + px, frame_number);
            pbr.bsdf = v3_make(1, 1, 1); pbr.emissive = v3_make(0, 0, 0);
            int missed = 0;
            int state = 0;   /* PlaygroundTraceState: 0 primitives pass, 1 Gaussians pass, 2 terminate (params.trace_state) */
            real T = 1; v3 rad = v3_make(0, 0, 0);   /* RayData: density = 1 - T, radiance */
            v3 lastO = rayOri, lastD = rayDir;
            while (!missed && (r_sqrt(v3_dot(thr, thr)) > R_(0.0001)) && accA < R_(0.995) && (pbr.pbr_bounces < max_pbr_bounces) && (numBounces < 32) &&
                   (state != 2)) {
                const v3 o = rayOri, d = rayDir;
                pbr.bsdf = v3_make(1, 1, 1); pbr.emissive = v3_make(0, 0, 0);
                /* traceMesh: closest triangle in (1e-5, 1e5) */
                state = 0;
                real best_t = R_(3.0e38), bu = 0, bv = 0; int best_f = -1;
                for (uint32_t f = 0; f < mesh->num_faces; ++f) {
                    real t, u, v;
                    if (tri_intersect(mesh, f, o, d, R_(1e-5), R_(1e5), &t, &u, &v) && t < best_t) { best_t = t; bu = u; bv = v; best_f = (int)f; }
                }
                real t_hit = 0;
                if (best_f < 0) missed = 1;   /* __miss__ms */
                else {   /* __closesthit__ch */
                    const int32_t* tr = mesh->triangles + 3 * (size_t)best_f;
                    pg_hit h; h.tri = (uint32_t)best_f; h.bu = bu; h.bv = bv;
                    v3 normal;
                    if (opts & 1u) {
                        const float* n0 = mesh->vertex_normals + 3 * (size_t)tr[0]; const float* n1 = mesh->vertex_normals + 3 * (size_t)tr[1];
                        const float* n2 = mesh->vertex_normals + 3 * (size_t)tr[2];
                        const real w0 = 1 - bu - bv;
                        normal = v3_make(w0 * n0[0] + bu * n1[0] + bv * n2[0], w0 * n0[1] + bu * n1[1] + bv * n2[1], w0 * n0[2] + bu * n1[2] + bv * n2[2]);
                        { const real len = r_sqrt(v3_dot(normal, normal)); normal = v3_make(normal.x / len, normal.y / len, normal.z / len); }   /* `/= length(...)` */
                    } else {
                        const float* p0 = mesh->vertices + 3 * (size_t)tr[0]; const float* p1 = mesh->vertices + 3 * (size_t)tr[1];
                        const float* p2 = mesh->vertices + 3 * (size_t)tr[2];
                        normal = v3_safe_normalize(v3_cross(v3_make(p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]), v3_make(p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2])));
                    }
                    real hit_t = best_t;
                    v3 new_dir = v3_make(0, 0, 0);
                    const int type = mesh->prim_type[best_f];
                    int next_state = 1;
                    if (type == 1) { new_dir = mirror_dir(d, normal); numBounces++; }
                    else if (type == 2) {
                        const real ior = (real)mesh->refractive_index[best_f] / R_(1.0003);
                        if (refract_dir(&new_dir, d, normal, ior)) hit_t += R_(1e-5);
                        else { new_dir = mirror_dir(d, normal); numBounces++; }
                    } else if (type == 3) {   /* handleDiffuse: the Gaussians in front of the surface, then the surface itself */
                        real T0 = T; v3 rad0 = rad;
                        if (!(opts & 2u)) { state = 1; trace_segment(cfg, N, density12, sph, sph_deg, min_T, inst12, scene6, o, d, R_(1e-9), hit_t, cands, &T, &rad); }
                        const v3 vrad = v3_sub(rad, rad0); const real valpha = (1 - T) - (1 - T0);
                        accC = v3_add(accC, vrad); accA += valpha;
                        const v3 dc = pg_diffuse_color(mesh, &h, opts, d, normal);
                        const real sa = 1 - accA;
                        accC = v3_add(accC, v3_scale(dc, sa)); accA += sa;
                        next_state = 2;
                    } else if (type == 4) {   /* handlePBR */
                        pg_cook_torrance(mesh, &h, opts, d, normal, px, py, frame_number, &pbr, &new_dir);
                        { const real len = r_sqrt(v3_dot(new_dir, new_dir)); new_dir = v3_make(new_dir.x / len, new_dir.y / len, new_dir.z / len); }
                    } else new_dir = d;
                    t_hit = hit_t;
                    rayOri = v3_add(o, v3_scale(d, hit_t)); rayDir = new_dir;
                    state = next_state;
                }
                /* the Gaussians between the ray origin and the surface (or the ray's end): traceGaussians sets the trace state itself */
                const real next_t = missed ? ray_t_max : t_hit;
                real T0 = T; v3 rad0 = rad;
                if (!(opts & 2u)) { state = 1; trace_segment(cfg, N, density12, sph, sph_deg, min_T, inst12, scene6, o, d, R_(1e-9), next_t, cands, &T, &rad); }
                const v3 radiance = v3_sub(rad, rad0); const real density = (1 - T) - (1 - T0);
                accA += density * (1 - accA);
                accC = v3_add(accC, v3_mul(thr, radiance));
                direct = v3_add(direct, radiance);
                thr = v3_scale(thr, 1 - density);
                accC = v3_add(accC, v3_mul(thr, v3_add(direct, pbr.emissive)));
                thr = v3_mul(thr, pbr.bsdf);
                lastO = o; lastD = d;
                if (++timeout > 1000) break;
            }
            direct = v3_add(direct, pg_background(mesh, lastD));
            thr = v3_scale(thr, 1 - accA);
            accC = v3_add(accC, v3_mul(thr, direct));
            accA = r_min(r_max(accA, 0), 1);
            out_rgba[4 * r] = accC.x; out_rgba[4 * r + 1] = accC.y; out_rgba[4 * r + 2] = accC.z; out_rgba[4 * r + 3] = accA;
            if (out_last_ray) { real* q = out_last_ray + 6 * (size_t)r; q[0] = lastO.x; q[1] = lastO.y; q[2] = lastO.z; q[3] = lastD.x; q[4] = lastD.y; q[5] = lastD.z; }
            if (out_bounces) out_bounces[r] = numBounces;
        }
        free(cands);
    }
    return 0;
}

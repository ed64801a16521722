// ref_grt_emul.inl — the OptiX traversal emulation shared by ref_grt_trace.cpp (forward programs) and ref_grt_trace_bwd.cpp
// (backward programs); included AFTER the reference's .cu, whose `params`, __intersection__is and __anyhit__ah it drives.
// This file is the part that is NOT the reference's (see the header of ref_grt_trace.cpp).  TEST INFRASTRUCTURE ONLY.
// the accessors have no constructors in a device compile (the host fills them): same layout, filled by copy
template <class A>
static void fill_accessor(A& acc, float* data, const int32_t* sizes, const int32_t* strides) {
    struct Raw { float* data; int32_t sizes[4]; int32_t strides[4]; } raw;
    raw.data = data;
    for (int i = 0; i < 4; ++i) { raw.sizes[i] = sizes[i]; raw.strides[i] = strides[i]; }
    static_assert(sizeof(Raw) == sizeof(A), "PackedTensorAccessor32<float, 4> layout");
    std::memcpy(&acc, &raw, sizeof(raw));
}

// ---- the emulated traversal ---------------------------------------------------------------------------------------------
namespace {
struct Scene {
    // Which far end the ray / box overlap test of the traversal uses once accepted hits have shrunk the ray: the CURRENT tmax
    // (what a traversal that happens to reach the box late would do) or the tmax the trace call started with (what one that
    // reaches it early would do).  OptiX leaves the order unspecified, so both are legal outcomes; they differ only for a
    // proxy whose hit distance precedes the ray's entry into its box.  The oracle takes the second (order-independent) one.
    bool box_test_uses_shrunk_tmax = false;
    float trace_tmax = 0.f;
    uint32_t n = 0;
    std::vector<float> inv;   // [n][12]: rows of the inverse linear part, then the translation of the instance transform
    // triangle-mesh proxies (render.primitive_type icosahedron ...: one triangle GAS over all particles' meshes, optixTracer.cpp:836-845)
    uint32_t num_triangles = 0;
    const float* vertices = nullptr;   // [V,3] world space, as the reference's mesh kernel wrote them
    const int32_t* triangles = nullptr;
    // custom primitives (render.primitive_type custom: one custom-primitive GAS over the particles' world boxes, optixTracer.cpp:638-655, 810-817)
    uint32_t num_boxes = 0;
    const float* boxes = nullptr;      // [n,6] min xyz, max xyz, as computeGaussianEnclosingAABBKernel wrote them
    // sphere proxies (render.primitive_type sphere: one built-in sphere GAS over the particles' enclosing spheres, optixTracer.cpp:765-781, 823-833)
    uint32_t num_spheres = 0;
    const float* centers = nullptr;    // [n,3] / [n]: as computeGaussianEnclosingSphereKernel wrote them (particlePrimitives.cu:386-403)
    const float* radii = nullptr;
    // Optional per-ray candidate subsets (ref_grt_set_ray_candidates): ray r is offered only the particles cand[cand_begin[r] .. cand_begin[r+1])
    // (ascending, so the index order of the full loops is kept).  The caller guarantees a CONSERVATIVE superset - every particle whose proxy
    // the ray's line can touch at all (tests/golden/make_fullsize_golden.py: bounding spheres in float64 with a margin) - so the programs
    // see exactly the reports the full loop would give them; it only makes 1 M-particle frames affordable for more than a handful of rays.
    const uint32_t* cand_begin = nullptr;
    const uint32_t* cand = nullptr;
    uint32_t cur_ray = 0;
    // optional per-ray log of what the traces returned (see optixTrace): ids / distances [rays, log_cap], counts [rays]
    uint32_t* log_ids = nullptr; float* log_ts = nullptr; uint32_t* log_num = nullptr; uint32_t log_cap = 0;
} g_scene;
// first / one-past-last position of the current ray's primitives in its candidate list, or the whole range [0, n)
static inline uint32_t cand_count(uint32_t n) { return g_scene.cand_begin ? g_scene.cand_begin[g_scene.cur_ray + 1] - g_scene.cand_begin[g_scene.cur_ray] : n; }
static inline uint32_t cand_at(uint32_t k) { return g_scene.cand_begin ? g_scene.cand[g_scene.cand_begin[g_scene.cur_ray] + k] : k; }
}  // namespace

bool optixReportIntersection(float t, unsigned) {
    if (!(t >= g_optix.tmin && t <= g_optix.tmax)) return false;
    const float far_end = g_optix.tmax;
    g_optix.tmax = t;          // the any-hit program sees the reported distance as the ray's tmax
    g_optix.ignore = false;
    __anyhit__ah();
    if (g_optix.ignore) { g_optix.tmax = far_end; return false; }
    return true;               // accepted: the ray now ends at t
}

static void optix_traverse(float3 o, float3 d, unsigned ray_flags);
void optixTrace(OptixTraversableHandle, float3 o, float3 d, float tmin, float tmax, float, OptixVisibilityMask, unsigned ray_flags, unsigned, unsigned,
                unsigned, uint32_t& p0, uint32_t& p1, uint32_t& p2, uint32_t& p3, uint32_t& p4, uint32_t& p5, uint32_t& p6, uint32_t& p7,
                uint32_t& p8, uint32_t& p9, uint32_t& p10, uint32_t& p11, uint32_t& p12, uint32_t& p13, uint32_t& p14, uint32_t& p15,
                uint32_t& p16, uint32_t& p17, uint32_t& p18, uint32_t& p19, uint32_t& p20, uint32_t& p21, uint32_t& p22, uint32_t& p23,
                uint32_t& p24, uint32_t& p25, uint32_t& p26, uint32_t& p27, uint32_t& p28, uint32_t& p29, uint32_t& p30, uint32_t& p31) {
    uint32_t* ps[32] = {&p0, &p1, &p2, &p3, &p4, &p5, &p6, &p7, &p8, &p9, &p10, &p11, &p12, &p13, &p14, &p15,
                        &p16, &p17, &p18, &p19, &p20, &p21, &p22, &p23, &p24, &p25, &p26, &p27, &p28, &p29, &p30, &p31};
    for (int k = 0; k < 32; ++k) g_optix.payload[k] = ps[k];
    g_optix.worldOrigin = o; g_optix.worldDirection = d;
    g_optix.tmin = tmin; g_optix.tmax = tmax;
    g_scene.trace_tmax = tmax;
    optix_traverse(o, d, ray_flags);
    // optional hit log (ref_grt_set_hit_log): what this trace returned to the raygen program - the (particle, distance) pairs of the payload,
    // nearest first - appended to the current ray's row.  The raygen program processes them in this order while the ray is above
    // min_transmittance; tests/golden/make_fullsize_golden.py stores the rows of the rays that differ from the checker, so that the GPU test can
    // say WHICH hits the reference's programs took in another order (or which particle they took instead)
    if (g_scene.log_ids) {
        const uint32_t r = g_scene.cur_ray;
        for (int k = 0; k < 16; ++k) {
            const uint32_t id = *g_optix.payload[2 * k];
            if (id == 0xFFFFFFFFu) break;
            uint32_t& n = g_scene.log_num[r];
            if (n < g_scene.log_cap) { g_scene.log_ids[(size_t)r * g_scene.log_cap + n] = id; std::memcpy(&g_scene.log_ts[(size_t)r * g_scene.log_cap + n], g_optix.payload[2 * k + 1], 4); }
            n++;
        }
    }
}
static void optix_traverse(float3 o, float3 d, unsigned ray_flags) {
#ifdef SHIM_OPTIX_TRIANGLE_PROXIES
    // Built-in triangles with OPTIX_RAY_FLAG_CULL_BACK_FACING_TRIANGLES (referenceOptix.cu:62): every FRONT-facing triangle (counter-clockwise
    // seen from the ray origin, OptiX's default) the ray crosses within its CURRENT interval is reported to the any-hit program, in
    // index order (OptiX leaves the order open; the payload ends up with the 16 nearest either way, up to ties).  Moeller-Trumbore in
    // world space, separately rounded fp32 operations.
    const uint32_t tpp = params.gPrimNumTri ? params.gPrimNumTri : 1u;   // a candidate subset names PARTICLES: all tpp triangles of each, in index order
    const uint32_t n_units = g_scene.cand_begin ? cand_count(0) * tpp : g_scene.num_triangles;
    for (uint32_t k = 0; k < n_units; ++k) {
        const uint32_t f = g_scene.cand_begin ? cand_at(k / tpp) * tpp + k % tpp : k;
        const int32_t* tri = g_scene.triangles + 3 * (size_t)f;
        const float* pa = g_scene.vertices + 3 * (size_t)tri[0];
        const float* pb = g_scene.vertices + 3 * (size_t)tri[1];
        const float* pc = g_scene.vertices + 3 * (size_t)tri[2];
        const float e1x = pb[0] - pa[0], e1y = pb[1] - pa[1], e1z = pb[2] - pa[2], e2x = pc[0] - pa[0], e2y = pc[1] - pa[1], e2z = pc[2] - pa[2];
        const float px = d.y * e2z - d.z * e2y, py = d.z * e2x - d.x * e2z, pz = d.x * e2y - d.y * e2x;
        const float det = e1x * px + e1y * py + e1z * pz;     // = d . (e2 x e1) = -(d . n): positive for a front face
        // back faces are culled when the trace asks for it (OPTIX_RAY_FLAG_CULL_BACK_FACING_TRIANGLES: every closed mesh; the trisurfel
        // pipelines trace with OPTIX_RAY_FLAG_NONE, referenceOptix.cu:62); parallel rays hit nothing
        if ((ray_flags & OPTIX_RAY_FLAG_CULL_BACK_FACING_TRIANGLES) ? !(det > 0.f) : !(det != 0.f)) continue;
        const float tx = o.x - pa[0], ty = o.y - pa[1], tz = o.z - pa[2];
        const float u = (tx * px + ty * py + tz * pz) / det;
        if (!(u >= 0.f && u <= 1.f)) continue;
        const float qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;
        const float v = (d.x * qx + d.y * qy + d.z * qz) / det;
        if (!(v >= 0.f && u + v <= 1.f)) continue;
        const float t = (e2x * qx + e2y * qy + e2z * qz) / det;
        // the ray interval is OPEN for built-in triangles: the raygen programs re-trace from tmin = (last hit distance + 1e-9), which is
        // the last hit distance itself in fp32 - a hit AT tmin reported again would never let the round loop advance
        if (!(t > g_optix.tmin && t < g_optix.tmax)) continue;
        g_optix.primitive = f;
        g_optix.barycentrics = float2{u, v};   // (read by the surfel pipeline's any-hit program only, barycentricSurfelsOptix.cu:210)
        optixReportIntersection(t, 0);
    }
    return;
#endif
#ifdef SHIM_OPTIX_SPHERE_PROXIES
    // Built-in spheres (OPTIX_PRIMITIVE_TYPE_SPHERE, the built-in intersection module of optixTracer.cpp:405-408, 452-454 - no program of the
    // reference's runs here): the intersector offers the any-hit program the ray's ENTRY into the sphere and, when that offer does not end
    // the ray (the program ignored it, or it lies outside the ray's interval), its EXIT - the second root.  (Back-face culling is a
    // triangle flag: it does not apply.)  __anyhit__ah ignores every offer but the one that fills the payload's last slot, so a ray is
    // offered the SAME particle twice - at both roots - and processes it twice when both fall into its rounds, like the trihexa's repeats.
    // NVIDIA does not publish the intersector's arithmetic; the emulation fixes it: roots of |po + t pd|^2 = 1 in the frame scaled by the
    // radius, po = (o - c) / r, pd = d / r, separately rounded fp32 operations, t = (-b -+ sqrt(b^2 - a (|po|^2 - 1))) / a; the interval
    // is open like the triangles' (a hit AT tmin reported again would never let the round loop advance).
    for (uint32_t k = 0, nk = cand_count(g_scene.num_spheres); k < nk; ++k) {
        const uint32_t i = cand_at(k);
        const float ir = 1.f / g_scene.radii[i];
        const float* c = g_scene.centers + 3 * (size_t)i;
        const float pox = (o.x - c[0]) * ir, poy = (o.y - c[1]) * ir, poz = (o.z - c[2]) * ir;
        const float pdx = d.x * ir, pdy = d.y * ir, pdz = d.z * ir;
        const float a = fmaf(pdz, pdz, fmaf(pdy, pdy, pdx * pdx)), b = fmaf(poz, pdz, fmaf(poy, pdy, pox * pdx));
        const float cc = fmaf(poz, poz, fmaf(poy, poy, pox * pox)) - 1.f;
        const float disc = fmaf(b, b, -(a * cc));
        if (!(disc >= 0.f) || !(a > 0.f)) continue;
        const float sq = sqrtf(disc);
        const float roots[2] = {(-b - sq) / a, (-b + sq) / a};
        for (int q = 0; q < 2; ++q) {
            const float t = roots[q];
            if (!(t > g_optix.tmin && t < g_optix.tmax)) continue;
            g_optix.primitive = i;
            if (optixReportIntersection(t, 0)) break;   // accepted: the ray ends at the entry, its exit lies beyond
        }
    }
    return;
#endif
#ifdef SHIM_OPTIX_CUSTOM_PROXIES
    // Custom primitives: the intersection program (intersectCustomParticle, world-space ray) runs for every particle whose WORLD box the
    // ray overlaps within its interval - the same slab test and the same far-end convention as for the instances' unit boxes below.
    // (OptiX may also call the program for a ray that misses the box narrowly; the emulation's boxes are exact.)
    for (uint32_t k = 0, nk = cand_count(g_scene.num_boxes); k < nk; ++k) {
        const uint32_t i = cand_at(k);
        const float* bx = g_scene.boxes + 6 * (size_t)i;
        const float ax0 = (bx[0] - o.x) / d.x, ax1 = (bx[3] - o.x) / d.x, ay0 = (bx[1] - o.y) / d.y, ay1 = (bx[4] - o.y) / d.y;
        const float az0 = (bx[2] - o.z) / d.z, az1 = (bx[5] - o.z) / d.z;
        const float tn = fmaxf(fmaxf(fminf(ax0, ax1), fminf(ay0, ay1)), fminf(az0, az1));
        const float tf = fminf(fminf(fmaxf(ax0, ax1), fmaxf(ay0, ay1)), fmaxf(az0, az1));
        const float far_end = g_scene.box_test_uses_shrunk_tmax ? g_optix.tmax : g_scene.trace_tmax;
        if (!(tn <= tf) || !(tf >= g_optix.tmin) || !(tn <= far_end)) continue;
        g_optix.primitive = i;
        __intersection__is();
    }
    return;
#endif
#ifndef SHIM_OPTIX_NO_INTERSECTION_PROGRAM   // (the surfel pipeline has triangles only: no __intersection__is to run)
    for (uint32_t k = 0, nk = cand_count(g_scene.n); k < nk; ++k) {
        const uint32_t i = cand_at(k);
        const float* m = &g_scene.inv[12 * (size_t)i];
        const float dx = o.x - m[9], dy = o.y - m[10], dz = o.z - m[11];
        const float3 oo = make_float3(m[0] * dx + m[1] * dy + m[2] * dz, m[3] * dx + m[4] * dy + m[5] * dz, m[6] * dx + m[7] * dy + m[8] * dz);
        const float3 od = make_float3(m[0] * d.x + m[1] * d.y + m[2] * d.z, m[3] * d.x + m[4] * d.y + m[5] * d.z, m[6] * d.x + m[7] * d.y + m[8] * d.z);
        // ray / unit box overlap within the ray's current interval (the hardware's job)
        const float ax0 = (-1.f - oo.x) / od.x, ax1 = (1.f - oo.x) / od.x, ay0 = (-1.f - oo.y) / od.y, ay1 = (1.f - oo.y) / od.y;
        const float az0 = (-1.f - oo.z) / od.z, az1 = (1.f - oo.z) / od.z;
        const float tn = fmaxf(fmaxf(fminf(ax0, ax1), fminf(ay0, ay1)), fminf(az0, az1));
        const float tf = fminf(fminf(fmaxf(ax0, ax1), fmaxf(ay0, ay1)), fmaxf(az0, az1));
        const float far_end = g_scene.box_test_uses_shrunk_tmax ? g_optix.tmax : g_scene.trace_tmax;
        if (!(tn <= tf) || !(tf >= g_optix.tmin) || !(tn <= far_end)) continue;
        g_optix.instance = i; g_optix.objectOrigin = oo; g_optix.objectDirection = od;
        __intersection__is();
    }
#endif
}

// instance matrices (object -> world, row-major 3x4, as the reference's instance kernel writes them) -> inverse maps
static void set_scene(uint32_t n, const float* transforms) {
    g_scene.n = n;
    g_scene.inv.resize(12 * (size_t)n);
    for (uint32_t i = 0; i < n; ++i) {   // inverse of the 3x3 linear part by cofactors (double), translation kept
        const float* t = &transforms[12 * (size_t)i];
        const double a = t[0], b = t[1], c = t[2], d = t[4], e = t[5], f = t[6], g = t[8], h = t[9], k = t[10];
        const double det = a * (e * k - f * h) - b * (d * k - f * g) + c * (d * h - e * g);
        const double inv[9] = {(e * k - f * h) / det, (c * h - b * k) / det, (b * f - c * e) / det, (f * g - d * k) / det, (a * k - c * g) / det,
                               (c * d - a * f) / det, (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det};
        float* m = &g_scene.inv[12 * (size_t)i];
        for (int q = 0; q < 9; ++q) m[q] = (float)inv[q];
        m[9] = t[3]; m[10] = t[7]; m[11] = t[11];
    }
}

// the members of PipelineParameters shared by the forward and the backward launch
static void set_common_params(int width, int height, const float* ray_to_world, const float* ray_o, const float* ray_d, const float* density12,
                              const float* sph48, const float* scene_aabb6, float min_transmittance, float min_response, float min_alpha,
                              unsigned sph_degree, float* features, float* density, float* hit_distance2, float* normals, float* hits_count,
                              int32_t* visibility) {
    static thread_local int32_t sz3[4], st3[4], sz1[4], st1[4], sz2[4], st2[4], szf[4], stf[4];
    const int32_t af[4] = {1, height, width, RAY_FEATURE_DIM}, bf[4] = {height * width * RAY_FEATURE_DIM, width * RAY_FEATURE_DIM, RAY_FEATURE_DIM, 1};
    for (int i = 0; i < 4; ++i) { szf[i] = af[i]; stf[i] = bf[i]; }
    const int32_t a3[4] = {1, height, width, 3}, b3[4] = {height * width * 3, width * 3, 3, 1};
    const int32_t a1[4] = {1, height, width, 1}, b1[4] = {height * width, width, 1, 1};
    const int32_t a2[4] = {1, height, width, 2}, b2[4] = {height * width * 2, width * 2, 2, 1};
    for (int i = 0; i < 4; ++i) { sz3[i] = a3[i]; st3[i] = b3[i]; sz1[i] = a1[i]; st1[i] = b1[i]; sz2[i] = a2[i]; st2[i] = b2[i]; }
    for (int r = 0; r < 3; ++r) params.rayToWorld[r] = make_float4(ray_to_world[4 * r], ray_to_world[4 * r + 1], ray_to_world[4 * r + 2], ray_to_world[4 * r + 3]);
    fill_accessor(params.rayOrigin, const_cast<float*>(ray_o), sz3, st3);
    fill_accessor(params.rayDirection, const_cast<float*>(ray_d), sz3, st3);
    params.particleDensity = reinterpret_cast<const ParticleDensity*>(density12);
    params.particleFeatures = sph48;
    params.particleExtendedData = nullptr;
    params.particleVisibility = visibility;
    fill_accessor(params.rayFeatures, features, szf, stf);
    fill_accessor(params.rayDensity, density, sz1, st1);
    fill_accessor(params.rayHitDistance, hit_distance2, sz2, st2);
    fill_accessor(params.rayNormal, normals, sz3, st3);
    fill_accessor(params.rayHitsCount, hits_count, sz1, st1);
    params.handle = 0;
    params.aabb = OptixAabb{scene_aabb6[0], scene_aabb6[1], scene_aabb6[2], scene_aabb6[3], scene_aabb6[4], scene_aabb6[5]};
    params.minTransmittance = min_transmittance;
    params.hitMinGaussianResponse = min_response;
    params.alphaMinThreshold = min_alpha;
    params.sphDegree = sph_degree;
    params.frameBounds = uint2{(unsigned)width - 1, (unsigned)height - 1};
    params.frameNumber = 0;
    params.gPrimNumTri = 0;
}
static void set_scene_triangles(uint32_t num_triangles, uint32_t triangles_per_particle, const float* vertices, const int32_t* triangles) {
    g_scene.num_triangles = num_triangles; g_scene.vertices = vertices; g_scene.triangles = triangles;
    params.gPrimNumTri = triangles_per_particle;
}

static void set_scene_boxes(uint32_t n, const float* boxes) { g_scene.num_boxes = n; g_scene.boxes = boxes; }
static void set_scene_spheres(uint32_t n, const float* centers, const float* radii) {
    g_scene.num_spheres = n; g_scene.centers = centers; g_scene.radii = radii;
    params.gPrimNumTri = 1;   // "number of primitives per gaussian" (optixTracer.cpp:766-767): optixPrimitiveIndex() divides by it
}

static void launch_raygen(int width, int height) {
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            g_optix.launchIndex = uint3{(unsigned)x, (unsigned)y, 0u};
            g_scene.cur_ray = (uint32_t)(y * width + x);
            __raygen__rg();
        }
}

// per-ray candidate subsets for the NEXT launches of this library: offsets [rays + 1], particles [offsets[rays]] ascending per ray; nulls clear
// per-ray hit log of the NEXT launches (rows zeroed by the caller); nulls switch it off
extern "C" void ref_grt_set_hit_log(uint32_t* ids, float* ts, uint32_t* num, uint32_t cap) { g_scene.log_ids = ids; g_scene.log_ts = ts; g_scene.log_num = num; g_scene.log_cap = cap; }
extern "C" void ref_grt_set_ray_candidates(const uint32_t* offsets, const uint32_t* particles) { g_scene.cand_begin = offsets; g_scene.cand = particles; }

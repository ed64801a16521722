// gut_render_nht.inl — neural harmonic features on the pixel-pair half-tile sweeps (included by gut_render.hip inside its anonymous
// namespace; shares RayPair, stage_entry, pair_geometry, the checkpoint / slot scheme and the task mapping of the SH sweeps).
//
// Reference behaviour (restated, not translated): gutKBufferRenderer.cuh:199-225, 273-352 (forward tile loop with per-ray features),
// :546-641 (evalBackwardNoKBuffer, per-ray-features branch), neuralHarmonicFeaturesParticle.slang:47-66, 117-127 (barycentric weights of the
// canonical tetrahedron), :146-196 (activation), :198-211 (integration), gaussianParticles.slang:181-190 (canonical intersection).
//
// The specialised configuration is the reference's default model (configs/base_gs.yaml:96-103): 48 feature floats per particle = 4
// tetrahedron vertices x 12, sincos activation with one frequency -> 24 ray features.  Other shapes keep the generic strip kernels of
// gut_render.hip.  What changed against those (47.6 ms per 1080p step at 1 M particles, profiles/r03zz_workloads):
//   * one wave64 per 16x8 half tile with a PAIR of pixels per lane: the 24 accumulators, the blend and the activation arguments are
//     packed fp32; entries are staged once per half tile (two waves per tile instead of four);
//   * the 48 feature floats of a round's entries are staged in LDS once, TRANSPOSED to [dimension][vertex], so that one 16-byte
//     broadcast read feeds the four-vertex blend of a dimension;
//   * the backward is a task per (256-entry segment, half tile) started from the forward's checkpoint of {T, D, 24 partial sums},
//     like the SH gradient sweep, instead of one wave per strip sweeping the whole list;
//   * geometric gradients leave through the entry's slot (16 words: B, d density, M, direct scale terms; contracted once per particle
//     by gut_grad_gather_kernel) - no atomics; the 48 feature-row words are summed over the wave through the LDS transposition and
//     leave as ONE 48-lane atomic instruction per (half tile, entry) on consecutive words of the particle's row.
constexpr uint32_t kNhtSegment = 256;              // the feature sweeps' checkpoint interval: a multiple of kGutSegment (their checkpoints are 26 KB per
                                                   // segment and half tile; the SH sweeps' 64-entry segments measured no faster here)
static_assert(kNhtSegment % kGutSegment == 0, "feature checkpoints sit on SH segment boundaries");
__host__ __device__ inline uint32_t nht_boundaries(uint32_t num_boundaries) { return (num_boundaries + kNhtSegment / kGutSegment - 1u) / (kNhtSegment / kGutSegment); }
constexpr uint32_t kNhtBatch = 32;                 // staged entries per round (LDS: 32 x (96 + 192) B per wave)
constexpr int kNhtIpd = 12, kNhtRay = 24, kNhtK = 48;
constexpr int kNhtCkQuads = 13;                    // checkpoint of a lane: {T, D} + 24 partial sums, for two pixels = 13 float4

// barycentric weights of the canonical tetrahedron as affine functions of the canonical point: w_k = gw_k . p + c_k
struct NhtTet {
    f3 gw0, gw1, gw2, gw3;
    float c0, c1, c2, c3;
};
__device__ __forceinline__ NhtTet nht_tet() {
    const float edge = 4.898979485566356f, face_h = 4.242640687119285f, face_in = 1.4142135623730951f;
    const f3 v0 = mk3(0.5f * edge, -face_in, -1.f), v1 = mk3(-0.5f * edge, -face_in, -1.f), v2 = mk3(0.f, face_h - face_in, -1.f), v3 = mk3(0.f, 0.f, 3.f);
    const f3 e1 = v1 - v0, e2 = v2 - v0, e3 = v3 - v0;
    const f3 c23 = cross(e2, e3);
    const float inv_det = 1.f / dot(e1, c23);
    NhtTet t;
    t.gw1 = c23 * inv_det; t.gw2 = cross(e3, e1) * inv_det; t.gw3 = cross(e1, e2) * inv_det;
    t.gw0 = (t.gw1 + t.gw2 + t.gw3) * -1.f;
    t.c1 = -dot(t.gw1, v0); t.c2 = -dot(t.gw2, v0); t.c3 = -dot(t.gw3, v0);
    t.c0 = 1.f - t.c1 - t.c2 - t.c3;
    return t;
}
__device__ __forceinline__ v2f nht_weight(f3 gw, float c, const p3& a) { return pfma(gw.x, a.x, pfma(gw.y, a.y, pfma(gw.z, a.z, splat(c)))); }
// (v_sin_f32 / v_cos_f32 take revolutions and return 0 outside [-256, 256]: the argument is reduced with v_fract_f32 first - exact, one
// instruction - so features that drift far from their initial [-pi/2, pi/2] during training keep their activation)
__device__ __forceinline__ v2f nht_sin2(v2f rev) { return v2f{__builtin_amdgcn_sinf(__builtin_amdgcn_fractf(rev.x)), __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(rev.y))}; }
__device__ __forceinline__ v2f nht_cos2(v2f rev) { return v2f{__builtin_amdgcn_cosf(__builtin_amdgcn_fractf(rev.x)), __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(rev.y))}; }
constexpr float kInvTwoPi = 0.15915494309189535f;

// Stages the feature rows of a round's entries: lane l takes the vertex pair (l & 1) of entry l >> 1 - 24 consecutive floats - and
// writes them transposed, s_feat[entry][dim] = {f0, f1, f2, f3}[dim].  idx = particle of entry l >> 1 (0xFFFFFFFF: padding, zeros).
__device__ __forceinline__ void nht_stage_features(const GutParams& P, const float* __restrict__ features, uint32_t idx, int lane,
                                                   float4* __restrict__ s_feat) {
    float v[24];
    const int hsel = lane & 1;
    if (idx == 0xFFFFFFFFu) {
#pragma unroll
        for (int i = 0; i < 24; ++i) v[i] = 0.f;
    } else if (P.sph_half) {
        const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const __half*>(features) + (size_t)idx * kNhtK + 24 * hsel);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const uint4 w = src[q];
            const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const __half2 h2 = *reinterpret_cast<const __half2*>(&ws[k]);
                v[8 * q + 2 * k] = __low2float(h2);
                v[8 * q + 2 * k + 1] = __high2float(h2);
            }
        }
    } else {
        const float4* src = reinterpret_cast<const float4*>(features + (size_t)idx * kNhtK + 24 * hsel);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const float4 w = src[q];
            v[4 * q] = w.x; v[4 * q + 1] = w.y; v[4 * q + 2] = w.z; v[4 * q + 3] = w.w;
        }
    }
    float2* dst = reinterpret_cast<float2*>(s_feat + (size_t)(lane >> 1) * kNhtIpd) + hsel;
#pragma unroll
    for (int m = 0; m < kNhtIpd; ++m) dst[2 * m] = make_float2(v[m], v[kNhtIpd + m]);
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
struct NhtFwdState {
    v2f T, D, cnt;
    v2f acc[kNhtRay];
};
template <int DEG, bool CKPT, bool UNI>
__device__ __forceinline__ void nht_fwd_sweep(const GutParams& P, const RayPair& rp, uint2 range, uint32_t half, int lane, const EntryLists& lists,
                                              const float4* __restrict__ density12, const float* __restrict__ features, float4* __restrict__ ck_nht,
                                              const GutCheckpoints& ck, float4* __restrict__ s_rec, float4* __restrict__ s_feat, NhtFwdState& st) {
    const NhtTet tet = nht_tet();
    bool alive0 = rp.valid0, alive1 = rp.valid1;
    v2f T = splat(1.f), D = splat(0.f), cnt = splat(0.f);
    v2f acc[kNhtRay];
#pragma unroll
    for (int i = 0; i < kNhtRay; ++i) acc[i] = splat(0.f);
    uint32_t b = range.x;
    RawEntry next = load_entry<false>(b + lane, min(range.y, (b & ~(kNhtBatch - 1u)) + kNhtBatch), lists, density12, nullptr);
    while (b < range.y) {
        if (!__any(alive0 || alive1)) break;
        const uint32_t bend = min(range.y, (b & ~(kNhtBatch - 1u)) + kNhtBatch);
        if (CKPT && b > range.x && (b % kNhtSegment) == 0) {
            float4* out = ck_nht + ((size_t)(b / kNhtSegment) * 2 + half) * (kNhtCkQuads * 64) + lane;
            out[0] = make_float4(alive0 ? T.x : 0.f, alive1 ? T.y : 0.f, D.x, D.y);   // dead pixels restart dead
#pragma unroll
            for (int q = 0; q < 12; ++q) out[64 * (q + 1)] = make_float4(acc[2 * q].x, acc[2 * q].y, acc[2 * q + 1].x, acc[2 * q + 1].y);
            if (lane == 0) ck.reached[(size_t)(b / kGutSegment) * 2 + half] = 1;
        }
        if (lane < (int)kNhtBatch) stage_entry<DEG, false>(P, next, UNI, rp.origin, &s_rec[lane * kRecQuads]);
        nht_stage_features(P, features, (uint32_t)__shfl((int)next.idx, lane >> 1, 64), lane, s_feat);
        __syncthreads();
        next = load_entry<false>(bend + lane, min(range.y, bend + kNhtBatch), lists, density12, nullptr);
        const int n = (int)(bend - b);
        for (int j = 0; j < n; ++j) {
            if (!__any(alive0 || alive1)) break;
            const float4* rec = &s_rec[j * kRecQuads];
            const PairGeom g = pair_geometry<UNI>(rp, rec);
            const bool c0 = g.acc0 && alive0, c1 = g.acc1 && alive1;
            if (!__any(c0 || c1)) continue;
            const float4 r3 = rec[3];
            const v2f il2 = prcp(g.l2);
            const v2f gray = g.cc * il2;
            const v2f resp = pair_response<DEG>(gray);
            const v2f ad = resp * r3.w;
            const v2f vu = pdot(g.v, g.u);
            const v2f t = vu * il2;
            // canonical intersection a = u - v (v.u)/|v|^2; hit distance |S v| |v.u| / |v|^2   (gaussianParticles.slang:181-190)
            const p3 a = p3{pfma(-t, g.v.x, g.u.x), pfma(-t, g.v.y, g.u.y), pfma(-t, g.v.z, g.u.z)};
            const p3 sv = p3{r3.x * g.v.x, r3.y * g.v.y, r3.z * g.v.z};
            const v2f ss = pdot(sv, sv) * (vu * vu);
            const v2f hitT = v2f{__builtin_amdgcn_sqrtf(ss.x), __builtin_amdgcn_sqrtf(ss.y)} * il2;
            const bool h0 = c0 && (hitT.x > rp.tmin.x) && (hitT.x < rp.tmax.x);
            const bool h1 = c1 && (hitT.y > rp.tmin.y) && (hitT.y < rp.tmax.y);
            const v2f alpha = psel(h0, h1, v2f{fminf(P.max_alpha, ad.x), fminf(P.max_alpha, ad.y)}, splat(0.f));
            const v2f hT = psel(h0, h1, hitT, splat(0.f));
            const v2f w = alpha * T;
            D = pfma(hT, w, D);
            T = T * (1.f - alpha);
            cnt += psel(w.x > 0.f, w.y > 0.f, splat(1.f), splat(0.f));
            alive0 = alive0 && !(T.x < P.min_transmittance);
            alive1 = alive1 && !(T.y < P.min_transmittance);
            // features: four-vertex blend at the canonical intersection, sin / cos, integrate with the hit's weight
            const v2f w0 = nht_weight(tet.gw0, tet.c0, a), w1 = nht_weight(tet.gw1, tet.c1, a), w2 = nht_weight(tet.gw2, tet.c2, a),
                      w3 = nht_weight(tet.gw3, tet.c3, a);
            const float4* fj = &s_feat[j * kNhtIpd];
#pragma unroll
            for (int m = 0; m < kNhtIpd; ++m) {
                const float4 F = fj[m];
                const v2f base = pfma(F.w, w3, pfma(F.z, w2, pfma(F.y, w1, F.x * w0)));
                const v2f rev = base * kInvTwoPi;
                acc[2 * m] = pfma(nht_sin2(rev), w, acc[2 * m]);
                acc[2 * m + 1] = pfma(nht_cos2(rev), w, acc[2 * m + 1]);
            }
        }
        __syncthreads();
        b = bend;
    }
    st.T = T; st.D = D; st.cnt = cnt;
#pragma unroll
    for (int i = 0; i < kNhtRay; ++i) st.acc[i] = acc[i];
}

template <int DEG, bool CKPT>
#ifndef GRUT_NHT_FWD_WAVES
#define GRUT_NHT_FWD_WAVES 4
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GRUT_NHT_FWD_WAVES, GRUT_NHT_FWD_WAVES)))
void gut_render_nhtp_fwd_kernel(GutParams P, const uint2* __restrict__ ranges, EntryLists lists, const float4* __restrict__ density12,
                                const float* __restrict__ features, const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                float* __restrict__ out_fd, float* __restrict__ out_dist, float* __restrict__ out_cnt, float4* __restrict__ ck_nht,
                                GutCheckpoints ck) {
    __shared__ float4 s_rec[kNhtBatch * kRecQuads];
    __shared__ float4 s_feat[kNhtBatch * kNhtIpd];
    uint32_t tile, half;
    half_mapping(blockIdx.x, tile, half);
    if (tile >= (uint32_t)(P.gx * P.gy)) return;
    tile = stride_permute(tile, (uint32_t)(P.gx * P.gy), 997);
    const int lane = threadIdx.x;
    const RayPair rp = init_ray_pair(P, ray_o, ray_d, tile, half, lane);
    const uint2 range = ranges[tile];
    NhtFwdState st;
    if (rp.uniform_origin) nht_fwd_sweep<DEG, CKPT, true>(P, rp, range, half, lane, lists, density12, features, ck_nht, ck, s_rec, s_feat, st);
    else nht_fwd_sweep<DEG, CKPT, false>(P, rp, range, half, lane, lists, density12, features, ck_nht, ck, s_rec, s_feat, st);
    constexpr size_t stride = kNhtRay + 1;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const bool inside = p ? rp.inside1 : rp.inside0, valid = p ? rp.valid1 : rp.valid0;
        if (!inside) continue;
        const size_t pix = (size_t)(p ? rp.py1 : rp.py0) * P.W + rp.px;
        if (P.out_half) {
            __half* o = reinterpret_cast<__half*>(out_fd) + pix * stride;
#pragma unroll
            for (int i = 0; i < kNhtRay; ++i) o[i] = __float2half(valid ? (p ? st.acc[i].y : st.acc[i].x) : 0.f);
            o[kNhtRay] = __float2half(valid ? 1.f - (p ? st.T.y : st.T.x) : 0.f);
        } else {
            float* o = out_fd + pix * stride;
#pragma unroll
            for (int i = 0; i < kNhtRay; ++i) o[i] = valid ? (p ? st.acc[i].y : st.acc[i].x) : 0.f;
            o[kNhtRay] = valid ? 1.f - (p ? st.T.y : st.T.x) : 0.f;
        }
        out_dist[pix] = valid ? (p ? st.D.y : st.D.x) : 1e6f;
        if (P.hitcounts) out_cnt[pix] = valid ? (p ? st.cnt.y : st.cnt.x) : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
// backward
//
// Per hit, with w = alpha T, T' = (1 - alpha) T, f = the hit's 24 activated features, Rem = C_fin - (partial sums up to and including
// this hit), resid = Rem / T' (what lies behind the hit; 0 when the ray ends here):
//   dL/d alpha = T sum_i (f_i - resid_i) gC_i + (hitT - resD) T gD - T_fin / (1 - alpha) gT          (no residual clamps: features are signed)
//   the gradient stops at an active alpha clamp (Slang reverse mode of min(max_alpha, .)): d response = density dalpha, d density =
//   response dalpha only while response * density < max_alpha
//   d base_m = w (gC_2m cos b_m - gC_2m+1 sin b_m);  d F[k][m] = w_k d base_m;  d w_k = sum_m F[k][m] d base_m;  gP = sum_k d w_k gw_k
// and with G = 2 (dL/d gray) a + gP the gradient w.r.t. the canonical intersection a = u - t v (t = v.u / |v|^2):
//   dL/du = G - v (v.G)/|v|^2,      dL/dv = -t G - u (v.G)/|v|^2 + 2 t v (v.G)/|v|^2
// (for the SH path G is a multiple of a, orthogonal to v, and the chain collapses; here it is general, as in the hit-distance terms).
// B = dL/du / scale, Bv = dL/dv / scale, M = B (x) (o - mu) + Bv (x) d leave through the slot like the SH sweep's terms.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float nht_wave_sum16(const float (&terms)[16], float* __restrict__ s_tr, int lane) {
#pragma unroll
    for (int k = 0; k < 16; ++k) s_tr[k * 65 + lane] = terms[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const float* col = &s_tr[(lane & 15) * 65 + (lane >> 4) * 16];
    float part = col[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) part += col[k];
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const v2u sa = __builtin_amdgcn_permlane32_swap(__float_as_uint(part), __float_as_uint(part), false, false);
    const float s2 = __uint_as_float(sa.x) + __uint_as_float(sa.y);
    const v2u sb = __builtin_amdgcn_permlane16_swap(__float_as_uint(s2), __float_as_uint(s2), false, false);
    const float tot = __uint_as_float(sb.x) + __uint_as_float(sb.y);
    __builtin_amdgcn_wave_barrier();   // the next pass overwrites s_tr
    return tot;   // every lane: wave total of term (lane & 15)
}

struct NhtGradIn {
    const float* fd;     // [H,W,25] upstream gradient of the packed image, or null:
    const float* feat;   // [H,W,24]
    const float* opa;    // [H,W,1]   (either may be null = zero)
};

template <int DEG, bool HAS_GDIST, bool UNI>
__device__ __forceinline__ void nht_bwd_sweep(const GutParams& P, const RayPair& rp, uint32_t seg_begin, uint32_t seg_end, int lane, uint32_t half,
                                              const EntryLists& lists, const float4* __restrict__ density12, const float* __restrict__ features,
                                              const GutGradSlots& slots, float* __restrict__ g_features, float4* __restrict__ s_rec,
                                              float4* __restrict__ s_feat, float* __restrict__ s_acc, float* __restrict__ s_tr, v2f T, v2f D,
                                              v2f Q, const v2f (&gC)[kNhtRay], v2f T_fin, v2f D_fin, v2f gT, v2f gD, bool alive0,
                                              bool alive1) {
    const NhtTet tet = nht_tet();
    v2f iT = prcp(T);
    uint32_t b = seg_begin;
    RawEntry next = load_entry(b + lane, min(seg_end, (b & ~(kNhtBatch - 1u)) + kNhtBatch), lists, density12, nullptr);
    while (b < seg_end) {
        if (!__any(alive0 || alive1)) break;
        const uint32_t bend = min(seg_end, (b & ~(kNhtBatch - 1u)) + kNhtBatch);
        if (lane < (int)kNhtBatch) {
            stage_entry<DEG, true>(P, next, UNI, rp.origin, &s_rec[lane * kRecQuads]);
            reinterpret_cast<uint32_t*>(&s_rec[lane * kRecQuads + 4])[0] = next.idx;   // (the radiance slot of the record is unused here)
        }
        nht_stage_features(P, features, (uint32_t)__shfl((int)next.idx, lane >> 1, 64), lane, s_feat);
        __syncthreads();
        next = load_entry(bend + lane, min(seg_end, bend + kNhtBatch), lists, density12, nullptr);
        const int n = (int)(bend - b);
        uint32_t hit_entries = 0u;
        for (int j = 0; j < n; ++j) {
            if (!__any(alive0 || alive1)) break;
            const float4* rec = &s_rec[j * kRecQuads];
            const PairGeom g = pair_geometry<UNI>(rp, rec);
            bool h0 = g.acc0 && alive0, h1 = g.acc1 && alive1;
            if (!__any(h0 || h1)) continue;
            const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
            const v2f il2 = prcp(g.l2);
            const v2f gray = g.cc * il2;
            const v2f resp = pair_response<DEG>(gray);
            const v2f ad = resp * r3.w;
            const v2f vu = pdot(g.v, g.u);
            const v2f t = vu * il2;
            const p3 a = p3{pfma(-t, g.v.x, g.u.x), pfma(-t, g.v.y, g.u.y), pfma(-t, g.v.z, g.u.z)};
            // hit distance (the forward's accept test includes its range): |S v| |v.u| / |v|^2 with S = 1 / r3.xyz here
            const p3 gscl = p3{prcp(splat(r3.x)), prcp(splat(r3.y)), prcp(splat(r3.z))};
            const p3 sv = p3{gscl.x * g.v.x, gscl.y * g.v.y, gscl.z * g.v.z};
            const v2f ss = pdot(sv, sv) * (vu * vu);
            const v2f hitT = v2f{__builtin_amdgcn_sqrtf(ss.x), __builtin_amdgcn_sqrtf(ss.y)} * il2;
            h0 = h0 && (hitT.x > rp.tmin.x) && (hitT.x < rp.tmax.x);
            h1 = h1 && (hitT.y > rp.tmin.y) && (hitT.y < rp.tmax.y);
            if (!__any(h0 || h1)) continue;
            hit_entries |= (1u << j);
            const v2f alpha = psel(h0, h1, v2f{fminf(P.max_alpha, ad.x), fminf(P.max_alpha, ad.y)}, splat(0.f));
            const v2f weight = alpha * T;
            const v2f oma = 1.f - alpha;
            const v2f nextT = oma * T;
            const v2f ioma = prcp(oma);
            const v2f inextT_raw = iT * ioma;
            const v2f inextT = psel(nextT.x <= P.min_transmittance, nextT.y <= P.min_transmittance, splat(0.f), inextT_raw);
            v2f dalpha = -(psel(alpha.x < 0.999999f, alpha.y < 0.999999f, T_fin * ioma, T) * gT);

            // features: blend, activation, un-blend; d base, d weights
            const v2f w0 = nht_weight(tet.gw0, tet.c0, a), w1 = nht_weight(tet.gw1, tet.c1, a), w2 = nht_weight(tet.gw2, tet.c2, a),
                      w3 = nht_weight(tet.gw3, tet.c3, a);
            const float4* fj = &s_feat[j * kNhtIpd];
            // three groups of four dimensions; after each group its 16 feature-row words (vertex k, dimension 4 c + ml -> word 4 k + ml of
            // the pass) are summed over the wave, so that d base lives for one group only
            v2f fsum = splat(0.f), dw0 = splat(0.f), dw1 = splat(0.f), dw2 = splat(0.f), dw3 = splat(0.f);
            float fw = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                v2f gb[4];
#pragma unroll
                for (int ml = 0; ml < 4; ++ml) {
                    const int m = 4 * c + ml;
                    const float4 F = fj[m];
                    const v2f base = pfma(F.w, w3, pfma(F.z, w2, pfma(F.y, w1, F.x * w0)));
                    const v2f rev = base * kInvTwoPi;
                    const v2f sn = nht_sin2(rev), cs = nht_cos2(rev);
                    fsum = pfma(sn, gC[2 * m], pfma(cs, gC[2 * m + 1], fsum));
                    gb[ml] = weight * pfma(cs, gC[2 * m], -(sn * gC[2 * m + 1]));
                    dw0 = pfma(F.x, gb[ml], dw0); dw1 = pfma(F.y, gb[ml], dw1); dw2 = pfma(F.z, gb[ml], dw2); dw3 = pfma(F.w, gb[ml], dw3);
                }
                float ft[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int k = q >> 2, ml = q & 3;
                    const v2f wk = k == 0 ? w0 : (k == 1 ? w1 : (k == 2 ? w2 : w3));
                    const v2f pr = wk * gb[ml];
                    ft[q] = pr.x + pr.y;
                }
                const float ftot = nht_wave_sum16(ft, s_tr, lane);
                if ((lane >> 4) == c) fw = ftot;
            }
            // sum_i (C_fin_i - partial sums_i) gC_i as ONE running scalar: Q -= w sum_i f_i gC_i (no residual clamps, so the channels need
            // not be kept apart - the 24 remainders would cost 48 registers and 48 packed operations per hit)
            Q = pfma(-weight, fsum, Q);
            dalpha = pfma(T, pfma(-inextT, Q, fsum), dalpha);
            const p3 gP = p3{pfma(tet.gw0.x, dw0, pfma(tet.gw1.x, dw1, pfma(tet.gw2.x, dw2, tet.gw3.x * dw3))),
                             pfma(tet.gw0.y, dw0, pfma(tet.gw1.y, dw1, pfma(tet.gw2.y, dw2, tet.gw3.y * dw3))),
                             pfma(tet.gw0.z, dw0, pfma(tet.gw1.z, dw1, pfma(tet.gw2.z, dw2, tet.gw3.z * dw3)))};

            p3 dl;
            if (UNI) dl = p3{splat(rp.origin.x - r0.w), splat(rp.origin.y - r1.w), splat(rp.origin.z - r2.w)};
            else dl = p3{rp.o.x - r0.w, rp.o.y - r1.w, rp.o.z - r2.w};

            p3 uX = p3{splat(0.f), splat(0.f), splat(0.f)}, vX = uX, sX = uX;   // hit-distance extras (gaussianParticles.cuh:545-580)
            if (HAS_GDIST) {
                const v2f il = v2f{__builtin_amdgcn_rsqf(g.l2.x), __builtin_amdgcn_rsqf(g.l2.y)};
                const p3 nrm = p3{g.v.x * il, g.v.y * il, g.v.z * il};
                const v2f pdt = -(vu * il);
                const p3 grdd = p3{nrm.x * pdt, nrm.y * pdt, nrm.z * pdt};
                const p3 grds = p3{gscl.x * grdd.x, gscl.y * grdd.y, gscl.z * grdd.z};
                const v2f gsq = pdot(grds, grds);
                const v2f gdist = v2f{__builtin_amdgcn_sqrtf(gsq.x), __builtin_amdgcn_sqrtf(gsq.y)};
                D = pfma(weight, gdist, D);
                const v2f resHitT = (D_fin - D) * inextT;
                dalpha = pfma((gdist - resHitT) * T, gD, dalpha);
                const v2f k = psel(gsq.x > 0.f, gsq.y > 0.f, weight * prcp(gdist) * gD, splat(0.f));
                const p3 grdsGrd = p3{grds.x * k, grds.y * k, grds.z * k};
                sX = p3{grdd.x * grdsGrd.x, grdd.y * grdsGrd.y, grdd.z * grdsGrd.z};
                const p3 sg = p3{gscl.x * grdsGrd.x, gscl.y * grdsGrd.y, gscl.z * grdsGrd.z};
                const v2f sd = pdot(sg, nrm);
                const p3 nGrd = p3{pfma(sg.x, pdt, -(g.u.x * sd)), pfma(sg.y, pdt, -(g.u.y * sd)), pfma(sg.z, pdt, -(g.u.z * sd))};
                uX = p3{-(nrm.x * sd), -(nrm.y * sd), -(nrm.z * sd)};
                const v2f ng = pdot(nrm, nGrd);
                vX = p3{(nGrd.x - nrm.x * ng) * il, (nGrd.y - nrm.y * ng) * il, (nGrd.z - nrm.z * ng) * il};
            }

            dalpha = psel(h0, h1, dalpha, splat(0.f));
            // the gradient stops at an active alpha clamp
            const bool open0 = ad.x < P.max_alpha, open1 = ad.y < P.max_alpha;
            const v2f dn = psel(open0, open1, resp * dalpha, splat(0.f));
            const v2f dresp = psel(open0, open1, r3.w * dalpha, splat(0.f));
            const v2f wg2 = psel(h0, h1, 2.f * v2f{particle_response_grd<DEG>(gray.x, resp.x, dresp.x),
                                                    particle_response_grd<DEG>(gray.y, resp.y, dresp.y)}, splat(0.f));
            const v2f mt = psel(h0, h1, splat(1.f), splat(0.f));
            const p3 G = p3{pfma(a.x, wg2, gP.x * mt), pfma(a.y, wg2, gP.y * mt), pfma(a.z, wg2, gP.z * mt)};
            const v2f vG = pdot(g.v, G) * il2;
            p3 uGrd = p3{pfma(-vG, g.v.x, G.x), pfma(-vG, g.v.y, G.y), pfma(-vG, g.v.z, G.z)};
            const v2f tvG2 = 2.f * t * vG;
            p3 vGrd = p3{pfma(tvG2, g.v.x, pfma(-vG, g.u.x, -(t * G.x))), pfma(tvG2, g.v.y, pfma(-vG, g.u.y, -(t * G.y))),
                         pfma(tvG2, g.v.z, pfma(-vG, g.u.z, -(t * G.z)))};
            if (HAS_GDIST) {
                uGrd = p3{pfma(uX.x, mt, uGrd.x), pfma(uX.y, mt, uGrd.y), pfma(uX.z, mt, uGrd.z)};
                vGrd = p3{pfma(vX.x, mt, vGrd.x), pfma(vX.y, mt, vGrd.y), pfma(vX.z, mt, vGrd.z)};
            }
            const p3 B = p3{r3.x * uGrd.x, r3.y * uGrd.y, r3.z * uGrd.z};
            const p3 Bv = p3{r3.x * vGrd.x, r3.y * vGrd.y, r3.z * vGrd.z};
            float terms[16];
            {
                v2f m[9];
                if (UNI) {   // (sum B) (x) (o - mu) is added at the flush
                    m[0] = Bv.x * rp.d.x; m[1] = Bv.x * rp.d.y; m[2] = Bv.x * rp.d.z; m[3] = Bv.y * rp.d.x; m[4] = Bv.y * rp.d.y; m[5] = Bv.y * rp.d.z;
                    m[6] = Bv.z * rp.d.x; m[7] = Bv.z * rp.d.y; m[8] = Bv.z * rp.d.z;
                } else {
                    m[0] = pfma(B.x, dl.x, Bv.x * rp.d.x); m[1] = pfma(B.x, dl.y, Bv.x * rp.d.y); m[2] = pfma(B.x, dl.z, Bv.x * rp.d.z);
                    m[3] = pfma(B.y, dl.x, Bv.y * rp.d.x); m[4] = pfma(B.y, dl.y, Bv.y * rp.d.y); m[5] = pfma(B.y, dl.z, Bv.y * rp.d.z);
                    m[6] = pfma(B.z, dl.x, Bv.z * rp.d.x); m[7] = pfma(B.z, dl.y, Bv.z * rp.d.y); m[8] = pfma(B.z, dl.z, Bv.z * rp.d.z);
                }
                terms[0] = B.x.x + B.x.y; terms[1] = B.y.x + B.y.y; terms[2] = B.z.x + B.z.y; terms[3] = dn.x + dn.y;
#pragma unroll
                for (int k = 0; k < 9; ++k) terms[4 + k] = m[k].x + m[k].y;
                const p3 sXm = p3{sX.x * mt, sX.y * mt, sX.z * mt};
                terms[13] = HAS_GDIST ? sXm.x.x + sXm.x.y : 0.f; terms[14] = HAS_GDIST ? sXm.y.x + sXm.y.y : 0.f;
                terms[15] = HAS_GDIST ? sXm.z.x + sXm.z.y : 0.f;
            }
            const float tot = nht_wave_sum16(terms, s_tr, lane);
            if (lane < 16) s_acc[j * 16 + lane] = tot;
            {
                // lane l < 48 holds the wave total of vertex (l & 15) >> 2, dimension 4 (l >> 4) + (l & 3)
                const uint32_t pid = reinterpret_cast<const uint32_t*>(&rec[4])[0];
                if (lane < kNhtK) atomicAdd(g_features + (size_t)pid * kNhtK + ((lane & 15) >> 2) * kNhtIpd + 4 * (lane >> 4) + (lane & 3), fw);
            }
            T = nextT;
            iT = inextT_raw;
            alive0 = alive0 && !(T.x < P.min_transmittance);
            alive1 = alive1 && !(T.y < P.min_transmittance);
        }
        __syncthreads();
        // flush: lane j stores the wave totals of staged entry j's 16 words to the entry's gradient slot
        if (lane < (int)kNhtBatch && ((hit_entries >> lane) & 1u)) {
            const float4* rec = &s_rec[lane * kRecQuads];
            const uint32_t pos = __float_as_uint(rec[5].w);
            const float4* accq = reinterpret_cast<const float4*>(&s_acc[lane * 16]);
            float4 a0 = accq[0], a1 = accq[1], a2 = accq[2], a3 = accq[3];
            if (UNI) {
                const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
                const f3 dl = rp.origin - mk3(r0.w, r1.w, r2.w);
                a1.x += a0.x * dl.x; a1.y += a0.x * dl.y; a1.z += a0.x * dl.z;
                a1.w += a0.y * dl.x; a2.x += a0.y * dl.y; a2.y += a0.y * dl.z;
                a2.z += a0.z * dl.x; a2.w += a0.z * dl.y; a3.x += a0.z * dl.z;
            }
            const size_t slot = 2 * (size_t)pos + half;
            float4* out = reinterpret_cast<float4*>(slots.partial + slot * 16);
            out[0] = a0; out[1] = a1; out[2] = a2; out[3] = a3;
            slots.flag[slot] = 1;
        }
        __syncthreads();
        b = bend;
    }
}

template <int DEG, bool HAS_GDIST>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gut_render_nhtp_bwd_kernel(GutParams P, const uint2* __restrict__ ranges, EntryLists lists, const float4* __restrict__ density12,
                                const float* __restrict__ features, const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                const float* __restrict__ fd, NhtGradIn g_in, const float* __restrict__ dist, const float* __restrict__ g_dist,
                                GutGradSlots slots, float* __restrict__ g_features, const float4* __restrict__ ck_nht, GutCheckpoints ck) {
    __shared__ float4 s_rec[kNhtBatch * kRecQuads];
    __shared__ float4 s_feat[kNhtBatch * kNhtIpd];
    __shared__ float s_tr[16 * 65];
    __shared__ float s_acc[kNhtBatch * 16];
    // task = (virtual tile, half), as gut_render_bwd_kernel: the segments that start at a checkpoint first, then the tiles' first segments
    // (the feature sweeps checkpoint every kNhtSegment entries: every kNhtSegment / kGutSegment-th boundary of the frame's tables)
    constexpr uint32_t kStep = kNhtSegment / kGutSegment;
    const uint32_t num_tiles = (uint32_t)(P.gx * P.gy), bnd_pad = (nht_boundaries(ck.num_boundaries) + 7u) & ~7u;
    uint32_t vtile, half;
    half_mapping(blockIdx.x, vtile, half);
    uint32_t tile, seg_begin, boundary = 0;
    bool from_checkpoint = false;
    if (vtile >= bnd_pad) {
        tile = vtile - bnd_pad;
        if (tile >= num_tiles) return;
        seg_begin = ranges[tile].x;
    } else {
        boundary = vtile * kStep;
        if (boundary == 0 || boundary >= ck.num_boundaries) return;
        if (!ck.reached[(size_t)boundary * 2 + half]) return;
        tile = ck.boundary_tile[boundary];
        if (tile >= num_tiles) return;
        seg_begin = boundary * kGutSegment;
        if (seg_begin <= ranges[tile].x) return;
        from_checkpoint = true;
    }
    const int lane = threadIdx.x;
    const uint32_t seg_end = min(ranges[tile].y, (seg_begin / kNhtSegment + 1u) * kNhtSegment);
    const RayPair rp = init_ray_pair(P, ray_o, ray_d, tile, half, lane);
    bool alive0 = rp.valid0, alive1 = rp.valid1;
    v2f T = splat(1.f), D = splat(0.f), T_fin = splat(0.f), D_fin = splat(0.f), gT = splat(0.f), gD = splat(0.f);
    v2f gC[kNhtRay], Q = splat(0.f);   // Q = sum_i (C_fin_i - partial sums_i) gC_i
#pragma unroll
    for (int i = 0; i < kNhtRay; ++i) gC[i] = splat(0.f);
    constexpr size_t stride = kNhtRay + 1;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (!(p ? alive1 : alive0)) continue;
        const size_t pix = (size_t)(p ? rp.py1 : rp.py0) * P.W + rp.px;
        float opa, gop, g[kNhtRay];
        if (g_in.fd) {
            const float* gq = g_in.fd + pix * stride;
#pragma unroll
            for (int i = 0; i < kNhtRay; ++i) g[i] = gq[i];
            gop = gq[kNhtRay];
        } else {
#pragma unroll
            for (int i = 0; i < kNhtRay; ++i) g[i] = g_in.feat ? g_in.feat[pix * kNhtRay + i] : 0.f;
            gop = g_in.opa ? g_in.opa[pix] : 0.f;
        }
        float q = 0.f;
        if (P.out_half) {
            const __half* f = reinterpret_cast<const __half*>(fd) + pix * stride;
#pragma unroll
            for (int i = 0; i < kNhtRay; ++i) q = fmaf(__half2float(f[i]), g[i], q);
            opa = __half2float(f[kNhtRay]);
        } else {
            const float* f = fd + pix * stride;
#pragma unroll
            for (int i = 0; i < kNhtRay; ++i) q = fmaf(f[i], g[i], q);
            opa = f[kNhtRay];
        }
#pragma unroll
        for (int i = 0; i < kNhtRay; ++i) { if (p) gC[i].y = g[i]; else gC[i].x = g[i]; }
        if (p) Q.y = q; else Q.x = q;
        if (p) { T_fin.y = 1.f - opa; gT.y = -gop; } else { T_fin.x = 1.f - opa; gT.x = -gop; }
        if (HAS_GDIST) {
            if (p) { D_fin.y = dist[pix]; gD.y = g_dist[pix]; } else { D_fin.x = dist[pix]; gD.x = g_dist[pix]; }
        }
    }
    if (from_checkpoint) {
        const float4* in = ck_nht + ((size_t)(boundary / kStep) * 2 + half) * (kNhtCkQuads * 64) + lane;
        const float4 c0 = in[0];
        T = v2f{c0.x, c0.y};
        if (HAS_GDIST) D = v2f{c0.z, c0.w};
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const float4 c = in[64 * (q + 1)];
            Q = pfma(-v2f{c.x, c.y}, gC[2 * q], pfma(-v2f{c.z, c.w}, gC[2 * q + 1], Q));
        }
        alive0 = alive0 && !(T.x < P.min_transmittance);
        alive1 = alive1 && !(T.y < P.min_transmittance);
    }
    if (rp.uniform_origin)
        nht_bwd_sweep<DEG, HAS_GDIST, true>(P, rp, seg_begin, seg_end, lane, half, lists, density12, features, slots, g_features, s_rec, s_feat, s_acc,
                                            s_tr, T, D, Q, gC, T_fin, D_fin, gT, gD, alive0, alive1);
    else
        nht_bwd_sweep<DEG, HAS_GDIST, false>(P, rp, seg_begin, seg_end, lane, half, lists, density12, features, slots, g_features, s_rec, s_feat, s_acc,
                                             s_tr, T, D, Q, gC, T_fin, D_fin, gT, gD, alive0, alive1);
}

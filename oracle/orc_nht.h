/*
 * orc_nht.h — the neural-harmonic-features model (kernels/slang/models/neuralHarmonicFeaturesParticle.slang) and one hit of the
 * Slang-pipeline backward, shared by the 3DGRT restatement (grt_oracle.c).  TEST INFRASTRUCTURE ONLY (see gut_oracle.c / grt_oracle.c).
 * nht = {particle_feature_dim K, interp_point_dim, support (0 centre, 1 tetrahedra), activation (0 none, 1 siren, 2 sincos, 3 relu),
 *        num_frequencies}.  The 3DGUT restatement (gut_oracle.c) carries its own copy of the same formulas; that copy is the one pinned
 * by the reference's render kernel (tests/golden/gut_nht.npz) and by float64 autograd (autograd_gut_nht.npz); tests compare the two.
 */
#ifndef ORC_NHT_H
#define ORC_NHT_H
#include "orc_math.h"

#define ORC_NHT_MAX_DIM 64

static int orc_nht_ray_dim(const int* nht) {
    return nht[3] == 2 ? nht[1] * nht[4] * 2 : ((nht[3] == 0 || nht[3] == 3) ? nht[1] : nht[1] * nht[4]);
}
/* canonical regular tetrahedron (:47-66) and the Cramer terms of its barycentric coordinates (:117-127) */
typedef struct { v3 v0, e1, e2, e3, c23; real inv_det; v3 gw[4]; } orc_nht_tet;
static orc_nht_tet orc_nht_tetra(void) {
    orc_nht_tet t;
    const real edge = R_(4.898979485566356), face_h = R_(4.242640687119285), face_in = R_(1.4142135623730951);
    const v3 v1 = v3_make(R_(-0.5) * edge, -face_in, R_(-1.0)), v2 = v3_make(0, face_h - face_in, R_(-1.0)), v3_ = v3_make(0, 0, R_(3.0));
    t.v0 = v3_make(R_(0.5) * edge, -face_in, R_(-1.0));
    t.e1 = v3_sub(v1, t.v0); t.e2 = v3_sub(v2, t.v0); t.e3 = v3_sub(v3_, t.v0);
    t.c23 = v3_cross(t.e2, t.e3);
    t.inv_det = 1 / v3_dot(t.e1, t.c23);
    t.gw[1] = v3_scale(t.c23, t.inv_det); t.gw[2] = v3_scale(v3_cross(t.e3, t.e1), t.inv_det); t.gw[3] = v3_scale(v3_cross(t.e1, t.e2), t.inv_det);
    t.gw[0] = v3_scale(v3_add(v3_add(t.gw[1], t.gw[2]), t.gw[3]), -1);
    return t;
}
static void orc_nht_weights(const int* nht, const orc_nht_tet* t, v3 P, real wq[4]) {
    wq[0] = 1; wq[1] = wq[2] = wq[3] = 0;
    if (nht[2] == 1) {
        const v3 d = v3_sub(P, t->v0);
        wq[1] = v3_dot(d, t->c23) * t->inv_det;
        wq[2] = v3_dot(t->e1, v3_cross(d, t->e3)) * t->inv_det;
        wq[3] = v3_dot(t->e1, v3_cross(t->e2, d)) * t->inv_det;
        wq[0] = 1 - wq[1] - wq[2] - wq[3];
    }
}
/* featuresFromParametersBuffer (:146-196): blended base features and the activated ray features */
static void orc_nht_features(const int* nht, const real* row, const real wq[4], real* base, real* out) {
    const int ipd = nht[1], act = nht[3], nf = nht[4], points = nht[2] == 1 ? 4 : 1;
    for (int n = 0; n < ipd; ++n) {
        base[n] = row[n] * wq[0];
        for (int k = 1; k < points; ++k) base[n] += wq[k] * row[k * ipd + n];
    }
    if (act == 0) { for (int i = 0; i < ipd; ++i) out[i] = base[i]; }
    else if (act == 3) { for (int i = 0; i < ipd; ++i) out[i] = r_max(0, base[i]); }
    else if (act == 2) {
        for (int k = 0; k < ipd; ++k)
            for (int f = 0; f < nf; ++f) {
                const real ang = base[k] * (real)(f + 1);
                out[k * nf * 2 + f * 2] = r_sin(ang); out[k * nf * 2 + f * 2 + 1] = r_cos(ang);
            }
    } else {
        for (int k = 0; k < ipd; ++k)
            for (int f = 0; f < nf; ++f) out[k * nf + f] = r_sin(base[k] * (real)ldexp(1.0, f));
    }
}
/* reverse of orc_nht_features: gf [ray_dim] -> g_row [K] (written), dP (returned) */
static v3 orc_nht_features_bwd(const int* nht, const orc_nht_tet* t, const real* row, const real wq[4], const real* base, const real* gf, real* g_row) {
    const int ipd = nht[1], act = nht[3], nf = nht[4], points = nht[2] == 1 ? 4 : 1;
    real gbase[ORC_NHT_MAX_DIM];
    for (int n = 0; n < ipd; ++n) gbase[n] = 0;
    if (act == 0) { for (int i = 0; i < ipd; ++i) gbase[i] = gf[i]; }
    else if (act == 3) { for (int i = 0; i < ipd; ++i) gbase[i] = base[i] > 0 ? gf[i] : 0; }
    else if (act == 2) {
        for (int k = 0; k < ipd; ++k)
            for (int q = 0; q < nf; ++q) {
                const real fr = (real)(q + 1), ang = base[k] * fr;
                gbase[k] += fr * (r_cos(ang) * gf[k * nf * 2 + q * 2] - r_sin(ang) * gf[k * nf * 2 + q * 2 + 1]);
            }
    } else {
        for (int k = 0; k < ipd; ++k)
            for (int q = 0; q < nf; ++q) {
                const real fr = (real)ldexp(1.0, q);
                gbase[k] += fr * r_cos(base[k] * fr) * gf[k * nf + q];
            }
    }
    v3 dP = v3_make(0, 0, 0);
    for (int k = 0; k < points; ++k) {
        real dwk = 0;
        for (int n = 0; n < ipd; ++n) {
            g_row[k * ipd + n] = wq[k] * gbase[n];
            dwk += row[k * ipd + n] * gbase[n];
        }
        if (points == 4) dP = v3_add(dP, v3_scale(t->gw[k], dwk));
    }
    return dP;
}
#endif

/*
 * ORACLE — test infrastructure only (included by sph2pob_oracle.c after the f32 / f64 instantiations).
 *
 * Unbiased IoU (Zhao et al., "Spherical criteria for fast and accurate 360 object detection"-style exact area of the
 * intersection of two spherical rectangles), restated from
 *     sphdet/iou/unbiased_iou_bfov.py:4-204   (BFoV,  class Sph)
 *     sphdet/iou/unbiased_iou_rbfov.py:4-182  (RBFoV, class Sph + roll_T :10-34)
 *     sphdet/iou/sph_iou_api.py:103-126       (unbiased_iou: expansion, jiter_spherical_bboxes, clamp)
 *
 * prec selects where float32 roundings happen:
 *   0  everything in double — what the reference computes when it is handed float64 tensors ("truth" fixtures);
 *   1  float32 spherical jitter and float32 deg2rad (sph_iou_api.py:121, unbiased_iou_bfov.py:189), then double —
 *      the definition the HIP kernel implements;
 *   2  the reference's own mixed arithmetic on float32 tensors: numpy keeps float32 wherever both operands are
 *      float32 (sin/cos of the inputs, V_lookat, V_up, N_up, N_down, the rotation matrix entries, the two areas)
 *      and is float64 elsewhere (V_right carries an np.zeros float64 column, :24-25).  sinf/cosf/acosf of libm stand
 *      in for numpy's float32 SIMD routines (both within 1 ulp, not bit-identical).
 */
#ifndef UNBIASED_IOU_ORACLE_H
#define UNBIASED_IOU_ORACLE_H

typedef struct { double v[3]; } ub_vec;

static inline double ub_f(double x, int prec) { return prec == 2 ? (double)(float)x : x; }
static inline double ub_sin(double x, int prec) { return prec == 2 ? (double)sinf((float)x) : sin(x); }
static inline double ub_cos(double x, int prec) { return prec == 2 ? (double)cosf((float)x) : cos(x); }

static inline ub_vec ub_cross(ub_vec a, ub_vec b) {
    ub_vec r = {{a.v[1] * b.v[2] - a.v[2] * b.v[1], a.v[2] * b.v[0] - a.v[0] * b.v[2], a.v[0] * b.v[1] - a.v[1] * b.v[0]}};
    return r;
}
static inline double ub_dot(ub_vec a, ub_vec b) { return a.v[0] * b.v[0] + a.v[1] * b.v[1] + a.v[2] * b.v[2]; }

/* roll_T (unbiased_iou_rbfov.py:10-34): rotate xyz about the unit axis n by gamma.  lo: operands are float32 arrays
 * in the reference (N_up / N_down), so every product and sum rounds to float32 under prec 2. */
static ub_vec ub_roll(ub_vec n, ub_vec p, double cg, double sg, int prec, int lo) {
    double m[3][3];
    double omc = ub_f(1.0 - cg, prec);
#define UBF(x) ub_f((x), prec)
    m[0][0] = UBF(UBF(UBF(n.v[0] * n.v[0]) * omc) + cg);
    m[0][1] = UBF(UBF(UBF(n.v[0] * n.v[1]) * omc) - UBF(n.v[2] * sg));
    m[0][2] = UBF(UBF(UBF(n.v[0] * n.v[2]) * omc) + UBF(n.v[1] * sg));
    m[1][0] = UBF(UBF(UBF(n.v[0] * n.v[1]) * omc) + UBF(n.v[2] * sg));
    m[1][1] = UBF(UBF(UBF(n.v[1] * n.v[1]) * omc) + cg);
    m[1][2] = UBF(UBF(UBF(n.v[1] * n.v[2]) * omc) - UBF(n.v[0] * sg));
    m[2][0] = UBF(UBF(UBF(n.v[0] * n.v[2]) * omc) - UBF(n.v[1] * sg));
    m[2][1] = UBF(UBF(UBF(n.v[1] * n.v[2]) * omc) + UBF(n.v[0] * sg));
    m[2][2] = UBF(UBF(UBF(n.v[2] * n.v[2]) * omc) + cg);
#undef UBF
    ub_vec r;
    int q = (prec == 2 && lo) ? 2 : 0;
    for (int i = 0; i < 3; i++)
        r.v[i] = ub_f(ub_f(ub_f(m[i][0] * p.v[0], q) + ub_f(m[i][1] * p.v[1], q), q) + ub_f(m[i][2] * p.v[2], q), q);
    return r;
}

/* getNormal (bfov :13-47, rbfov :48-86): N = [left, right, up, down], corner points V and their edge pairs. */
static void ub_normals(const double* b, int dim, int prec, ub_vec N[4], ub_vec V[4], int E[4][2]) {
    double th = b[0], ph = b[1], a2 = ub_f(b[2] / 2, prec), b2 = ub_f(b[3] / 2, prec);
    double st = ub_sin(th, prec), ct = ub_cos(th, prec), sp = ub_sin(ph, prec), cp = ub_cos(ph, prec);
    ub_vec look = {{ub_f(sp * ct, prec), ub_f(sp * st, prec), cp}};
    ub_vec right = {{-st, ct, 0.0}};
    ub_vec up = {{ub_f(-cp * ct, prec), ub_f(-cp * st, prec), sp}};
    double ca = ub_cos(a2, prec), sa = ub_sin(a2, prec), cb = ub_cos(b2, prec), sb = ub_sin(b2, prec);
    for (int k = 0; k < 3; k++) {
        /* left/right: float32 * float64 V_right -> float64; the sin * V_lookat product is float32 */
        N[0].v[k] = -ca * right.v[k] + ub_f(sa * look.v[k], prec);
        N[1].v[k] = ca * right.v[k] + ub_f(sa * look.v[k], prec);
        /* up/down: all float32 */
        N[2].v[k] = ub_f(ub_f(-cb * up.v[k], prec) + ub_f(sb * look.v[k], prec), prec);
        N[3].v[k] = ub_f(ub_f(cb * up.v[k], prec) + ub_f(sb * look.v[k], prec), prec);
    }
    if (dim == 5) {
        double cg = ub_cos(b[4], prec), sg = ub_sin(b[4], prec);
        for (int k = 0; k < 4; k++) N[k] = ub_roll(look, N[k], cg, sg, prec, k >= 2);
    }
    static const int e[4][2] = {{0, 2}, {3, 0}, {2, 1}, {1, 3}}; /* [left,up] [down,left] [up,right] [right,down] */
    for (int k = 0; k < 4; k++) {
        ub_vec c = ub_cross(N[e[k][0]], N[e[k][1]]);
        double nrm = sqrt(ub_dot(c, c));
        for (int j = 0; j < 3; j++) V[k].v[j] = c.v[j] / nrm;
        E[k][0] = e[k][0];
        E[k][1] = e[k][1];
    }
}

static inline double ub_angle(ub_vec a, ub_vec b) { /* interArea :49-53 */
    double c = -ub_dot(a, b);
    c = c < -1 ? -1 : (c > 1 ? 1 : c);
    return acos(c);
}
/* np.round(x, 8) >= 0 */
static inline int ub_nonneg(double d) { return rint(d * 1e8) / 1e8 >= 0; }

static double ub_area(double fx, double fy, int prec) { /* area :10-12 */
    if (prec == 2) {
        float s = -sinf((float)fx / 2) * sinf((float)fy / 2);
        return (double)(4 * acosf(s) - (float)(2 * 3.141592653589793));
    }
    return 4 * acos(-sin(fx / 2) * sin(fy / 2)) - 2 * 3.141592653589793;
}

static double unbiased_pair_iou(const double* in1, const double* in2, int dim, int prec) {
    double b1[5] = {0, 0, 0, 0, 0}, b2[5] = {0, 0, 0, 0, 0};
    if (prec == 0) {
        for (int k = 0; k < dim; k++) { b1[k] = in1[k]; b2[k] = in2[k]; }
        jitter_spherical_f64(b1, b2, dim);
        for (int k = 0; k < dim; k++) { b1[k] = deg2rad_f64(b1[k]); b2[k] = deg2rad_f64(b2[k]); }
    } else {
        float f1[5], f2[5];
        for (int k = 0; k < dim; k++) { f1[k] = (float)in1[k]; f2[k] = (float)in2[k]; }
        jitter_spherical_f32(f1, f2, dim);
        for (int k = 0; k < dim; k++) { b1[k] = (double)deg2rad_f32(f1[k]); b2[k] = (double)deg2rad_f32(f2[k]); }
    }
    ub_vec N[8], V[8];
    int E1[4][2], E2[4][2];
    ub_normals(b1, dim, prec, N, V, E1);
    ub_normals(b2, dim, prec, N + 4, V + 4, E2);
    int count = 0;
    double sum = 0;
    /* remove_outer_points (bfov :105-139): 8 corners, then 16 plane pairs x 2 antipodal points */
    for (int k = 0; k < 8; k++) {
        int inside = 1;
        for (int j = 0; j < 8; j++) inside &= ub_nonneg(ub_dot(V[k], N[j]));
        if (inside) {
            const int (*E)[2] = k < 4 ? E1 : E2;
            int base = k < 4 ? 0 : 4;
            sum += ub_angle(N[base + E[k & 3][0]], N[base + E[k & 3][1]]);
            count++;
        }
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            ub_vec t = ub_cross(N[i], N[4 + j]);
            double nrm = sqrt(ub_dot(t, t)) + 1e-10;
            ub_vec p = {{t.v[0] / nrm, t.v[1] / nrm, t.v[2] / nrm}};
            ub_vec t2 = ub_cross(N[4 + j], N[i]);
            double nrm2 = sqrt(ub_dot(t2, t2)) + 1e-10;
            ub_vec q = {{t2.v[0] / nrm2, t2.v[1] / nrm2, t2.v[2] / nrm2}};
            int in_p = 1, in_q = 1;
            for (int k = 0; k < 8; k++) {
                in_p &= ub_nonneg(ub_dot(p, N[k]));
                in_q &= ub_nonneg(ub_dot(q, N[k]));
            }
            if (in_p) { sum += ub_angle(N[i], N[4 + j]); count++; }
            if (in_q) { sum += ub_angle(N[4 + j], N[i]); count++; }
        }
    double inter = count ? sum - (count - 2) * 3.141592653589793 : 0.0; /* interArea :54-62 */
    double aa = ub_area(b1[2], b1[3], prec), ab = ub_area(b2[2], b2[3], prec);
    double au = prec == 2 ? (double)((float)aa + (float)ab) : aa + ab;
    const double eps = 1e-8;
    double iou = dim == 4 ? (inter + eps) / (au - (inter + eps))  /* bfov :200 */
                          : inter / (au - inter + eps);            /* rbfov :178 */
    float f = (float)iou;                                          /* .float(), then clamp(0, 1) sph_iou_api.py:126 */
    if (f != f) return (double)f;
    return f < 0 ? 0.0 : (f > 1 ? 1.0 : (double)f);
}

int sph2pob_oracle_unbiased_iou(const double* b1, const double* b2, double* out, int64_t n, int dim, int prec,
                                int nthreads) {
    if (dim != 4 && dim != 5) return -2;
    if (prec < 0 || prec > 2) return -3;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
#endif
    for (int64_t i = 0; i < n; i++) out[i] = unbiased_pair_iou(b1 + i * dim, b2 + i * dim, dim, prec);
    (void)nthreads;
    return 0;
}

#endif

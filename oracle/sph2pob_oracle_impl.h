/*
 * ORACLE (test infrastructure, not product): scalar CPU restatement of the reference's Sph2Pob hot path.
 * This file is a "template": it is included twice by sph2pob_oracle.c, once with REAL=float (the
 * reference's fp32 arithmetic, operation by operation, no FMA contraction) and once with REAL=double
 * (the same algorithm as accuracy truth; equals the reference run under torch.set_default_dtype(float64)).
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.  The planar
 * rotated-IoU kernel of mmcv-full 1.6.0 (mmcv/ops/csrc/common/box_iou_rotated_utils.hpp, un-vendored
 * dependency pinned at README.md:95 / docker/Dockerfile:17) is restated from its published algorithm in
 * FN(planar_inter_mmcv).
 */

#ifndef REAL
#error "include from sph2pob_oracle.c"
#endif

typedef struct { REAL x, y, w, h, a; } FN(pbox);
typedef struct { REAL x, y, z; } FN(v3);

#define R_(c) ((REAL)(c))

/* torch.deg2rad: self * M_PI_180 with the scalar cast to the tensor dtype (ATen UnaryOps). */
static inline REAL FN(deg2rad)(REAL x) { return x * R_(0.017453292519943295769236907684886127134428718885417); }

static inline REAL FN(rsin)(REAL x) { return (REAL)MSIN(x); }
static inline REAL FN(rcos)(REAL x) { return (REAL)MCOS(x); }

static inline REAL FN(clampr)(REAL x, REAL lo, REAL hi) {
    /* torch.clamp: min(max(x, lo), hi), NaN propagates */
    if (x != x) return x;
    REAL y = x < lo ? lo : x;
    return y > hi ? hi : y;
}

static inline FN(v3) FN(v3_make)(REAL x, REAL y, REAL z) { FN(v3) r = {x, y, z}; return r; }
static inline FN(v3) FN(v3_add)(FN(v3) a, FN(v3) b) { return FN(v3_make)(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline FN(v3) FN(v3_sub)(FN(v3) a, FN(v3) b) { return FN(v3_make)(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline REAL FN(v3_dot)(FN(v3) a, FN(v3) b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* torch.cross */
static inline FN(v3) FN(v3_cross)(FN(v3) a, FN(v3) b) {
    return FN(v3_make)(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* F.normalize(v, dim=1): v / max(||v||_2, 1e-12) */
static inline FN(v3) FN(v3_normalize)(FN(v3) a) {
    REAL n = (REAL)MSQRT((a.x * a.x + a.y * a.y) + a.z * a.z);
    REAL d = n < R_(1e-12) ? R_(1e-12) : n;
    return FN(v3_make)(a.x / d, a.y / d, a.z / d);
}

/* compute_angle_between_direction, radians.
 * sphdet/iou/sph2pob_efficient.py:192-208 (rad) ; sph2pob_standard.py:202-217 (returns rad/pi*180). */
static inline REAL FN(angle_between_rad)(FN(v3) a, FN(v3) b) {
    FN(v3) an = FN(v3_normalize)(a), bn = FN(v3_normalize)(b);
    REAL c = FN(clampr)(FN(v3_dot)(an, bn), R_(-1 + 1e-7), R_(1 - 1e-7));
    REAL r = (REAL)MACOS(c);
    return r < 0 ? -r : r;
}
static inline REAL FN(rad2deg_ref)(REAL r) { return r / R_(3.141592653589793) * R_(180); }

/* compute_clockwise_or_anticlockwise_between_direction: +1 iff (a x b) . ref < 0 else -1
 * sphdet/iou/sph2pob_efficient.py:211-226 ; sph2pob_standard.py:220-235 */
static inline REAL FN(sign_mask)(FN(v3) a, FN(v3) b, FN(v3) ref) {
    REAL s = FN(v3_dot)(FN(v3_cross)(a, b), ref);
    return s < 0 ? R_(1) : R_(-1);
}

/* compute_edge_length: sph2pob_standard.py:110-118 (same in efficient/legacy) */
static inline REAL FN(edge_length)(REAL fov, int edge) {
    if (edge == SPH2POB_EDGE_ARC) return fov;
    if (edge == SPH2POB_EDGE_TANGENT) return R_(2) * (REAL)MTAN(fov / R_(2));
    return R_(2) * FN(rsin)(fov / R_(2));
}

/* ------------------------------------------------------------------------------------------------ */
/* jiter_spherical_bboxes: sphdet/iou/sph_iou_api.py:244-260.  b1/b2 are writable copies (dim 4|5).   */
static void FN(jitter_spherical)(REAL* b1, REAL* b2, int dim) {
    const double eps = 1e-4 * 1.2345678;
    int similar = 0;
    for (int k = 0; k < dim; k++) {
        REAL d = b1[k] - b2[k];
        d = d < 0 ? -d : d;
        if (d < R_(eps)) similar = 1;
    }
    if (similar) {
        for (int k = 0; k < dim; k++) {
            b1[k] = b1[k] - R_(2 * eps);
            b2[k] = b2[k] + R_(eps);
        }
    }
    const double pi = 180;
    b1[0] = FN(clampr)(b1[0], R_(2 * eps), R_(2 * pi - eps));
    for (int k = 1; k < 4; k++) b1[k] = FN(clampr)(b1[k], R_(2 * eps), R_(pi - eps));
    b2[0] = FN(clampr)(b2[0], R_(eps), R_(2 * pi - 2 * eps));
    for (int k = 1; k < 4; k++) b2[k] = FN(clampr)(b2[k], R_(eps), R_(pi - 2 * eps));
    if (dim == 5) { /* quirk kept: only bboxes2's gamma is clamped (twice), :256-258 */
        b2[4] = FN(clampr)(b2[4], R_(-2 * pi + eps), R_(2 * pi - 2 * eps));
        b2[4] = FN(clampr)(b2[4], R_(-2 * pi + 2 * eps), R_(2 * pi - eps));
    }
}

/* jiter_rotated_bboxes: sphdet/iou/sph_iou_api.py:222-242 */
static void FN(jitter_rotated)(FN(pbox)* p1, FN(pbox)* p2) {
    double eps = 1e-4 * 1.2345678;
    REAL* b1 = &p1->x;
    REAL* b2 = &p2->x;
    const double e1[5] = {eps, eps, 2 * eps, 2 * eps, eps};
    const double e2[5] = {2 * eps, 2 * eps, eps, eps, 5 * eps};
    static const int cols[4] = {0, 2, 3, 4};
    int similar = 0;
    for (int c = 0; c < 4; c++) {
        REAL d = b1[cols[c]] - b2[cols[c]];
        d = d < 0 ? -d : d;
        if (d < R_(eps)) similar = 1;
    }
    if (similar) {
        for (int k = 0; k < 5; k++) {
            b1[k] = b1[k] + R_(e1[k]);
            b2[k] = b2[k] + R_(e2[k]);
        }
    }
    eps = 1e-3 * 1.2345678;
    REAL da = b1[4] - b2[4];
    da = da < 0 ? -da : da;
    if (da < R_(eps)) {
        b1[4] = b1[4] + R_(eps);
        b2[4] = b2[4] + R_(2 * eps);
    }
    const double pi = 3.141592653589793;
    for (int k = 2; k < 4; k++) {
        if (b1[k] < R_(2 * eps / 10)) b1[k] = R_(2 * eps / 10);
        if (b2[k] < R_(eps / 10)) b2[k] = R_(eps / 10);
    }
    b1[4] = FN(clampr)(b1[4], R_(-2 * pi + 2 * eps), R_(2 * pi - eps));
    b2[4] = FN(clampr)(b2[4], R_(-2 * pi + eps), R_(2 * pi - 2 * eps));
}

/* ------------------------------------------------------------------------------------------------ */
/* shared front end: deg2rad, sin/cos, 3-D centre c and meridian tangent d.
 * sph2pob_standard.py:23-41,121-172 ; sph2pob_efficient.py:29-46,110-160 */
typedef struct { REAL th, ph, al, be, ga; REAL st, ct, sp, cp; FN(v3) c, d; } FN(sbox);

static FN(sbox) FN(load_sbox)(const REAL* b, int dim) {
    FN(sbox) s;
    s.th = FN(deg2rad)(b[0]); s.ph = FN(deg2rad)(b[1]); s.al = FN(deg2rad)(b[2]); s.be = FN(deg2rad)(b[3]);
    s.ga = dim == 5 ? FN(deg2rad)(b[4]) : R_(0);
    s.st = FN(rsin)(s.th); s.ct = FN(rcos)(s.th); s.sp = FN(rsin)(s.ph); s.cp = FN(rcos)(s.ph);
    s.c = FN(v3_make)(s.sp * s.ct, s.sp * s.st, s.cp);
    s.d = FN(v3_make)(s.cp * s.ct, s.cp * s.st, -s.sp);
    return s;
}

/* compute_rotate_matrix(theta, phi): rows look, down, right.  sph2pob_standard.py:239-261 */
static void FN(rotate_matrix)(REAL th, REAL ph, REAL R[3][3]) {
    REAL st = FN(rsin)(th), ct = FN(rcos)(th), sp = FN(rsin)(ph), cp = FN(rcos)(ph);
    R[0][0] = sp * ct; R[0][1] = sp * st; R[0][2] = cp;
    R[1][0] = cp * ct; R[1][1] = cp * st; R[1][2] = -sp;
    R[2][0] = st;      R[2][1] = -ct;     R[2][2] = R_(0);
}
static inline FN(v3) FN(mat_vec)(REAL M[3][3], FN(v3) v) { /* torch.bmm (3x3)(3x1) */
    return FN(v3_make)((M[0][0] * v.x + M[0][1] * v.y) + M[0][2] * v.z,
                       (M[1][0] * v.x + M[1][1] * v.y) + M[1][2] * v.z,
                       (M[2][0] * v.x + M[2][1] * v.y) + M[2][2] * v.z);
}
static void FN(mat_mul)(REAL A[3][3], REAL B[3][3], REAL C[3][3]) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[i][j] = (A[i][0] * B[0][j] + A[i][1] * B[1][j]) + A[i][2] * B[2][j];
}
/* compute_gamma_matrix(theta, phi, gamma) = T^T (Rx(gamma) T).  sph2pob_standard.py:300-314 */
static void FN(gamma_matrix)(REAL th, REAL ph, REAL gamma, REAL out[3][3]) {
    REAL T[3][3], Rx[3][3], RT[3][3], Tt[3][3];
    FN(rotate_matrix)(th, ph, T);
    REAL sg = FN(rsin)(gamma), cg = FN(rcos)(gamma);
    Rx[0][0] = 1; Rx[0][1] = 0;  Rx[0][2] = 0;
    Rx[1][0] = 0; Rx[1][1] = cg; Rx[1][2] = -sg;
    Rx[2][0] = 0; Rx[2][1] = sg; Rx[2][2] = cg;
    FN(mat_mul)(Rx, T, RT);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Tt[i][j] = T[j][i];
    FN(mat_mul)(Tt, RT, out);
}

/* sph2pob_standard: sphdet/iou/sph2pob_standard.py:8-80 (output angle in radians, as the IoU API and the
 * loss call it: rbb_angle_version='rad'). */
static void FN(transform_standard)(const REAL* g_, const REAL* p_, int dim, int edge, int angle,
                                   FN(pbox)* og, FN(pbox)* op) {
    FN(sbox) g = FN(load_sbox)(g_, dim), p = FN(load_sbox)(p_, dim);
    REAL th_r = (g.th + p.th) / R_(2), ph_r = (g.ph + p.ph) / R_(2);
    REAL Rm[3][3];
    /* compute_rotate_matrix_auto :286-297 */
    FN(v3) df = FN(v3_sub)(g.c, p.c);
    REAL l1 = ((df.x < 0 ? -df.x : df.x) + (df.y < 0 ? -df.y : df.y)) + (df.z < 0 ? -df.z : df.z);
    if (l1 > R_(1e-8)) { /* compute_rotate_matrix_better :264-283 */
        FN(v3) look = FN(v3_normalize)(FN(v3_add)(g.c, p.c));
        FN(v3) right = FN(v3_normalize)(FN(v3_sub)(p.c, g.c));
        FN(v3) up = FN(v3_cross)(look, right);
        Rm[0][0] = look.x;  Rm[0][1] = look.y;  Rm[0][2] = look.z;
        Rm[1][0] = right.x; Rm[1][1] = right.y; Rm[1][2] = right.z;
        Rm[2][0] = up.x;    Rm[2][1] = up.y;    Rm[2][2] = up.z;
    } else {
        FN(rotate_matrix)(th_r, ph_r, Rm);
    }
    FN(v3) dg = g.d, dp = p.d;
    if (dim == 5) { /* :47-54 */
        REAL G[3][3];
        FN(gamma_matrix)(g.th, g.ph, -g.ga, G); dg = FN(mat_vec)(G, dg);
        FN(gamma_matrix)(p.th, p.ph, -p.ga, G); dp = FN(mat_vec)(G, dp);
    }
    FN(v3) cg = FN(mat_vec)(Rm, g.c), cp = FN(mat_vec)(Rm, p.c);
    dg = FN(mat_vec)(Rm, dg); dp = FN(mat_vec)(Rm, dp);

    const FN(v3) ez = {0, 0, 1}, ex = {1, 0, 0}, nez = {-R_(0), -R_(0), -1};
    FN(v3) dd[2] = {dg, dp}, cc[2] = {cg, cp};
    REAL ang_out[2], th_out[2], ph_out[2];
    for (int k = 0; k < 2; k++) {
        /* compute_internal_angle :88-108 */
        FN(v3) d = dd[k];
        if (angle == SPH2POB_ANGLE_PROJECT) d.x = 0;
        REAL a = FN(rad2deg_ref)(FN(angle_between_rad)(d, ez));
        a = a < 0 ? -a : a;
        a = a * FN(sign_mask)(ez, d, ex);
        ang_out[k] = FN(deg2rad)(a); /* standardize_rotated_box('rad') :342-364 */
        /* compute_spherical_coordinate :175-199 */
        FN(v3) c = cc[k];
        REAL ph = FN(rad2deg_ref)(FN(angle_between_rad)(c, ez));
        FN(v3) cxy = FN(v3_make)(c.x, c.y, 0);
        REAL th = FN(rad2deg_ref)(FN(angle_between_rad)(cxy, ex));
        th = th * FN(sign_mask)(ex, cxy, nez);
        th_out[k] = FN(deg2rad)(th);
        ph_out[k] = FN(deg2rad)(ph);
    }
    og->x = th_out[0]; og->y = ph_out[0]; og->w = FN(edge_length)(g.al, edge); og->h = FN(edge_length)(g.be, edge); og->a = ang_out[0];
    op->x = th_out[1]; op->y = ph_out[1]; op->w = FN(edge_length)(p.al, edge); op->h = FN(edge_length)(p.be, edge); op->a = ang_out[1];
}

/* sph2pob_efficient: sphdet/iou/sph2pob_efficient.py:9-73 */
static void FN(transform_efficient)(const REAL* g_, const REAL* p_, int dim, int edge, int angle,
                                    FN(pbox)* og, FN(pbox)* op) {
    FN(sbox) g = FN(load_sbox)(g_, dim), p = FN(load_sbox)(p_, dim);
    FN(v3) z = FN(v3_cross)(g.c, p.c);
    FN(v3) s = FN(v3_add)(g.c, p.c);
    FN(v3) ref = FN(v3_make)(s.x / R_(2), s.y / R_(2), s.z / R_(2));
    REAL arc = FN(angle_between_rad)(g.c, p.c);
    FN(v3) dg = g.d, dp = p.d;
    if (angle == SPH2POB_ANGLE_PROJECT) { dg.x = 0; dp.x = 0; } /* :92-93 (unrotated frame quirk) */
    REAL ag = FN(angle_between_rad)(dg, z) * FN(sign_mask)(z, dg, ref);
    REAL ap = FN(angle_between_rad)(dp, z) * FN(sign_mask)(z, dp, ref);
    if (dim == 5) { ag = ag - g.ga; ap = ap - p.ga; } /* :55-57 */
    og->x = 0;   og->y = 0; og->w = FN(edge_length)(g.al, edge); og->h = FN(edge_length)(g.be, edge); og->a = ag;
    op->x = arc; op->y = 0; op->w = FN(edge_length)(p.al, edge); op->h = FN(edge_length)(p.be, edge); op->a = ap;
}

/* sph2pob_legacy: sphdet/iou/sph2pob_legacy.py:8-31 (BFoV only; the caller rejects dim 5 like the
 * reference's 4-way torch.chunk does at :52-53). */
static REAL FN(legacy_angle_aux)(REAL th_box, REAL ph_box, REAL th_ref, REAL ph_ref) { /* :120-134 */
    REAL sb = FN(rsin)(th_box), cb = FN(rcos)(th_box), spb = FN(rsin)(ph_box), cpb = FN(rcos)(ph_box);
    REAL sr = FN(rsin)(th_ref), cr = FN(rcos)(th_ref), spr = FN(rsin)(ph_ref), cpr = FN(rcos)(ph_ref);
    FN(v3) db = FN(v3_make)(cpb * cb, cpb * sb, -spb);
    FN(v3) dr = FN(v3_make)(cpr * cr, cpr * sr, -spr);
    REAL a = FN(rad2deg_ref)(FN(angle_between_rad)(db, dr));
    a = a < 0 ? -a : a;
    const REAL hp = R_(3.141592653589793 / 2);
    int sign = ((th_box >= th_ref) && (ph_box < hp)) || ((th_box <= th_ref) && (ph_box > hp));
    if (!sign) a = a * R_(-1);
    return a;
}
static void FN(transform_legacy)(const REAL* g_, const REAL* p_, int dim, int edge, int angle,
                                 FN(pbox)* og, FN(pbox)* op) {
    (void)dim; (void)angle;
    REAL g[4] = {g_[0], g_[1], g_[2], g_[3]}, p[4] = {p_[0], p_[1], p_[2], p_[3]};
    /* standardize_spherical_box :236-257 */
    REAL dt = g[0] - p[0];
    dt = dt < 0 ? -dt : dt;
    if (dt > R_(180)) {
        g[0] = (REAL)MFMOD(g[0] + R_(180), R_(360));
        p[0] = (REAL)MFMOD(p[0] + R_(180), R_(360));
    }
    /* transform_position :38-83, 'convention' radians :217-234 */
    const REAL pi = R_(3.141592653589793), hpi = R_(3.141592653589793 / 2);
    REAL thg = FN(deg2rad)(g[0]) - pi, phg = hpi - FN(deg2rad)(g[1]);
    REAL thp = FN(deg2rad)(p[0]) - pi, php = hpi - FN(deg2rad)(p[1]);
    REAL phi_i = (phg + php) / R_(2);
    REAL phg_ = phg - phi_i, php_ = php - phi_i;
    REAL dphi = phg - php;   dphi = dphi < 0 ? -dphi : dphi;
    REAL dth = thg - thp;    dth = dth < 0 ? -dth : dth;
    REAL s1 = FN(rsin)(dphi / R_(2)), s2 = FN(rsin)(dth / R_(2));
    REAL L = R_(2) * (REAL)MASIN((REAL)MSQRT(s1 * s1 + (FN(rcos)(phg) * FN(rcos)(php)) * (s2 * s2)));
    REAL sl = FN(rsin)(L / R_(2)), sd = FN(rsin)(dphi / R_(2));
    REAL q = (sl * sl - sd * sd) / (FN(rcos)(phg_) * FN(rcos)(php_));
    REAL dth_ = R_(2) * (REAL)MASIN((REAL)MSQRT(q));
    dth_ = dth_ < 0 ? -dth_ : dth_;
    REAL sgn = (thp > thg) ? R_(1) : R_(-1);
    /* transfrom_anlge :102-118 ('math' radians) */
    REAL mg = FN(deg2rad)(g[0]), mp = FN(deg2rad)(p[0]);
    REAL pg = FN(deg2rad)(g[1]), pp = FN(deg2rad)(p[1]);
    REAL mid = (mg + mp) / R_(2);
    REAL ag = FN(legacy_angle_aux)(mg, pg, mid, pg);
    REAL ap = FN(legacy_angle_aux)(mp, pp, mid, pp);
    og->x = 0;          og->y = phg_;
    op->x = dth_ * sgn; op->y = php_;
    og->w = FN(edge_length)(FN(deg2rad)(g[2]), edge); og->h = FN(edge_length)(FN(deg2rad)(g[3]), edge);
    op->w = FN(edge_length)(FN(deg2rad)(p[2]), edge); op->h = FN(edge_length)(FN(deg2rad)(p[3]), edge);
    og->a = FN(deg2rad)(ag); op->a = FN(deg2rad)(ap);
}

static void FN(transform_dispatch)(int variant, const REAL* g, const REAL* p, int dim, int edge, int angle,
                                   FN(pbox)* og, FN(pbox)* op) {
    if (variant == SPH2POB_VARIANT_STANDARD) FN(transform_standard)(g, p, dim, edge, angle, og, op);
    else if (variant == SPH2POB_VARIANT_EFFICIENT) FN(transform_efficient)(g, p, dim, edge, angle, og, op);
    else FN(transform_legacy)(g, p, dim, edge, angle, og, op);
}

/* ------------------------------------------------------------------------------------------------ */
/* Planar rotated-rect intersection, mmcv-full 1.6.0 box_iou_rotated_utils.hpp (CPU flavour, T = REAL). */
typedef struct { REAL x, y; } FN(pt);
static inline REAL FN(cross2)(FN(pt) a, FN(pt) b) { return a.x * b.y - b.x * a.y; }
static inline REAL FN(dot2)(FN(pt) a, FN(pt) b) { return a.x * b.x + a.y * b.y; }
static inline FN(pt) FN(pt_sub)(FN(pt) a, FN(pt) b) { FN(pt) r = {a.x - b.x, a.y - b.y}; return r; }

static void FN(mmcv_vertices)(const FN(pbox)* b, FN(pt) pts[4]) {
    double theta = b->a;
    REAL c2 = (REAL)cos(theta) * 0.5f, s2 = (REAL)sin(theta) * 0.5f;
    pts[0].x = b->x - s2 * b->h - c2 * b->w;
    pts[0].y = b->y + c2 * b->h - s2 * b->w;
    pts[1].x = b->x + s2 * b->h - c2 * b->w;
    pts[1].y = b->y - c2 * b->h - s2 * b->w;
    pts[2].x = 2 * b->x - pts[0].x;
    pts[2].y = 2 * b->y - pts[0].y;
    pts[3].x = 2 * b->x - pts[1].x;
    pts[3].y = 2 * b->y - pts[1].y;
}

static int FN(mmcv_cmp)(const void* pa, const void* pb) {
    const FN(pt)* A = (const FN(pt)*)pa; const FN(pt)* B = (const FN(pt)*)pb;
    REAL t = FN(cross2)(*A, *B);
    int less_ab, less_ba;
    if (fabs((double)t) < 1e-6) {
        less_ab = FN(dot2)(*A, *A) < FN(dot2)(*B, *B);
        less_ba = FN(dot2)(*B, *B) < FN(dot2)(*A, *A);
    } else {
        less_ab = t > 0;
        less_ba = t < 0;
    }
    return less_ab ? -1 : (less_ba ? 1 : 0);
}

static REAL FN(planar_inter_mmcv)(const FN(pbox)* b1, const FN(pbox)* b2) {
    FN(pt) p1[4], p2[4], v1[4], v2[4], ip[24], q[24];
    FN(mmcv_vertices)(b1, p1);
    FN(mmcv_vertices)(b2, p2);
    for (int i = 0; i < 4; i++) { v1[i] = FN(pt_sub)(p1[(i + 1) % 4], p1[i]); v2[i] = FN(pt_sub)(p2[(i + 1) % 4], p2[i]); }
    int num = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            REAL det = FN(cross2)(v2[j], v1[i]);
            if (fabs((double)det) <= 1e-14) continue;
            FN(pt) v12 = FN(pt_sub)(p2[j], p1[i]);
            REAL t1 = FN(cross2)(v2[j], v12) / det;
            REAL t2 = FN(cross2)(v1[i], v12) / det;
            if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f) {
                ip[num].x = p1[i].x + v1[i].x * t1;
                ip[num].y = p1[i].y + v1[i].y * t1;
                num++;
            }
        }
    {
        FN(pt) AB = v2[0], DA = v2[3];
        REAL ABAB = FN(dot2)(AB, AB), ADAD = FN(dot2)(DA, DA);
        for (int i = 0; i < 4; i++) {
            FN(pt) AP = FN(pt_sub)(p1[i], p2[0]);
            REAL pab = FN(dot2)(AP, AB), pad = -FN(dot2)(AP, DA);
            if (pab >= 0 && pad >= 0 && pab <= ABAB && pad <= ADAD) ip[num++] = p1[i];
        }
    }
    {
        FN(pt) AB = v1[0], DA = v1[3];
        REAL ABAB = FN(dot2)(AB, AB), ADAD = FN(dot2)(DA, DA);
        for (int i = 0; i < 4; i++) {
            FN(pt) AP = FN(pt_sub)(p2[i], p1[0]);
            REAL pab = FN(dot2)(AP, AB), pad = -FN(dot2)(AP, DA);
            if (pab >= 0 && pad >= 0 && pab <= ABAB && pad <= ADAD) ip[num++] = p2[i];
        }
    }
    if (num <= 2) return 0;
    /* convex_hull_graham(shift_to_zero = true) */
    int t = 0;
    for (int i = 1; i < num; i++)
        if (ip[i].y < ip[t].y || (ip[i].y == ip[t].y && ip[i].x < ip[t].x)) t = i;
    FN(pt) start = ip[t];
    for (int i = 0; i < num; i++) q[i] = FN(pt_sub)(ip[i], start);
    FN(pt) tmp = q[0]; q[0] = q[t]; q[t] = tmp;
    qsort(q + 1, (size_t)(num - 1), sizeof(FN(pt)), FN(mmcv_cmp));
    REAL dist[24];
    for (int i = 0; i < num; i++) dist[i] = FN(dot2)(q[i], q[i]);
    int k;
    for (k = 1; k < num; k++) if (dist[k] > 1e-8) break;
    if (k == num) return 0;
    q[1] = q[k];
    int m = 2;
    for (int i = k + 1; i < num; i++) {
        while (m > 1 && FN(cross2)(FN(pt_sub)(q[i], q[m - 2]), FN(pt_sub)(q[m - 1], q[m - 2])) >= 0) m--;
        q[m++] = q[i];
    }
    if (m <= 2) return 0;
    REAL area = 0;
    for (int i = 1; i < m - 1; i++) {
        REAL c = FN(cross2)(FN(pt_sub)(q[i], q[0]), FN(pt_sub)(q[i + 1], q[0]));
        area += (REAL)fabs((double)c);
    }
    return (REAL)(area / 2.0);
}

/* single_box_iou_rotated (mmcv 1.6.0): centre shift, area guards, iou / iof. */
static REAL FN(planar_iou_mmcv)(const FN(pbox)* r1, const FN(pbox)* r2, int mode) {
    FN(pbox) b1 = *r1, b2 = *r2;
    double sx = (r1->x + r2->x) / 2.0, sy = (r1->y + r2->y) / 2.0;
    b1.x = (REAL)(r1->x - sx); b1.y = (REAL)(r1->y - sy);
    b2.x = (REAL)(r2->x - sx); b2.y = (REAL)(r2->y - sy);
    REAL a1 = b1.w * b1.h, a2 = b2.w * b2.h;
    if (a1 < 1e-14 || a2 < 1e-14) return 0;
    REAL inter = FN(planar_inter_mmcv)(&b1, &b2);
    REAL base = mode == SPH2POB_MODE_IOU ? (a1 + a2 - inter) : a1;
    return inter / base;
}

/* ------------------------------------------------------------------------------------------------ */
/* Vendored differentiable planar IoU: sphdet/iou/diff_iou_rotated.py:20-343 (forward value only).     */
static void FN(diff_corners)(const FN(pbox)* b, FN(pt) c[4]) { /* box2corners :297-322 */
    static const double fx[4] = {0.5, -0.5, -0.5, 0.5}, fy[4] = {0.5, 0.5, -0.5, -0.5};
    REAL s = FN(rsin)(b->a), co = FN(rcos)(b->a);
    for (int i = 0; i < 4; i++) {
        REAL x4 = R_(fx[i]) * b->w, y4 = R_(fy[i]) * b->h;
        REAL rx = x4 * co + y4 * (-s);
        REAL ry = x4 * s + y4 * co;
        c[i].x = rx + b->x;
        c[i].y = ry + b->y;
    }
}
static void FN(diff_in_box)(const FN(pt) c1[4], const FN(pt) c2[4], int in[4]) { /* box1_in_box2 :63-89 */
    FN(pt) a = c2[0], ab = FN(pt_sub)(c2[1], a), ad = FN(pt_sub)(c2[3], a);
    REAL nab = FN(dot2)(ab, ab), nad = FN(dot2)(ad, ad);
    for (int i = 0; i < 4; i++) {
        FN(pt) am = FN(pt_sub)(c1[i], a);
        REAL pab = FN(dot2)(ab, am), pad = FN(dot2)(ad, am);
        in[i] = (pab >= 0 && pab <= nab) && (pad >= 0 && pad <= nad);
    }
}
typedef struct { REAL ang; int idx; } FN(angidx);
static int FN(ang_cmp)(const void* a, const void* b) {
    const FN(angidx)* A = (const FN(angidx)*)a; const FN(angidx)* B = (const FN(angidx)*)b;
    if (A->ang < B->ang) return -1;
    if (A->ang > B->ang) return 1;
    return A->idx - B->idx;
}
static REAL FN(planar_inter_diff)(const FN(pbox)* b1, const FN(pbox)* b2) {
    FN(pt) c1[4], c2[4], vert[24];
    int mask[24];
    FN(diff_corners)(b1, c1);
    FN(diff_corners)(b2, c2);
    /* box_intersection :20-60 */
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            REAL x1 = c1[i].x, y1 = c1[i].y, x2 = c1[(i + 1) % 4].x, y2 = c1[(i + 1) % 4].y;
            REAL x3 = c2[j].x, y3 = c2[j].y, x4 = c2[(j + 1) % 4].x, y4 = c2[(j + 1) % 4].y;
            REAL num = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4);
            REAL dt = (x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4);
            REAL t = dt / num;
            if (num == 0) t = -1;
            int mt = (t > 0) && (t < 1);
            REAL du = (x1 - x2) * (y1 - y3) - (y1 - y2) * (x1 - x3);
            REAL u = -du / num;
            if (num == 0) u = -1;
            int mu = (u > 0) && (u < 1);
            int m = mt && mu;
            t = dt / (num + R_(1e-8));
            int k = 8 + i * 4 + j;
            vert[k].x = (x1 + t * (x2 - x1)) * (REAL)m;
            vert[k].y = (y1 + t * (y2 - y1)) * (REAL)m;
            if (!m) { vert[k].x = 0; vert[k].y = 0; } /* NaN*0 cannot occur for the masked-out reference rows either: keep 0 */
            mask[k] = m;
        }
    int in12[4], in21[4];
    FN(diff_in_box)(c1, c2, in12);
    FN(diff_in_box)(c2, c1, in21);
    /* check_overlap :196-223 */
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            if (c1[i].x == c2[j].x && c1[i].y == c2[j].y) { in12[i] = 1; in21[j] = 0; }
    for (int i = 0; i < 4; i++) { vert[i] = c1[i]; mask[i] = in12[i]; vert[4 + i] = c2[i]; mask[4 + i] = in21[i]; }
    /* sort_vertices :141-175, sort_normalized_vertices :226-256 */
    int nv = 0;
    REAL sx = 0, sy = 0;
    for (int k = 0; k < 24; k++) { nv += mask[k]; sx += vert[k].x * (REAL)mask[k]; sy += vert[k].y * (REAL)mask[k]; }
    if (nv < 3) return 0;
    REAL mx = sx / (REAL)nv, my = sy / (REAL)nv;
    FN(angidx) ai[24];
    for (int k = 0; k < 24; k++) {
        REAL x = vert[k].x - mx, y = vert[k].y - my;
        if (!mask[k]) { x = R_(-1e6); y = R_(1e-6); }
        ai[k].ang = (REAL)MATAN2(y, x);
        ai[k].idx = k;
    }
    qsort(ai, 24, sizeof(FN(angidx)), FN(ang_cmp));
    int index[25];
    for (int k = 0; k < 24; k++) index[k] = ai[k].idx;
    if (nv < 24) index[nv] = index[0];
    FN(pt) sel[9];
    for (int k = 0; k < 9; k++) {
        int valid = k < nv + 1;
        sel[k].x = valid ? vert[index[k]].x : 0;
        sel[k].y = valid ? vert[index[k]].y : 0;
    }
    /* calculate_area :178-193 */
    REAL total = 0;
    for (int k = 0; k < 8; k++) total += sel[k].x * sel[k + 1].y - sel[k].y * sel[k + 1].x;
    total = total < 0 ? -total : total;
    return total / R_(2);
}
static REAL FN(planar_iou_diff)(const FN(pbox)* b1, const FN(pbox)* b2, int mode) { /* :325-343 */
    REAL inter = FN(planar_inter_diff)(b1, b2);
    REAL a1 = b1->w * b1->h, a2 = b2->w * b2->h;
    return mode == SPH2POB_MODE_IOU ? inter / (a1 + a2 - inter) : inter / a1;
}

/* Independent exact intersection area (Sutherland-Hodgman clip of rect 1 by the 4 half-planes of rect 2,
 * shoelace about the first clipped vertex).  Not in the reference: it is the oracle's tolerance-free ground
 * truth used to rank the two reference planar algorithms (mmcv's hull has absolute 1e-6/1e-8 tolerances,
 * the vendored diff version mis-sorts some fp32 cases) when they disagree. */
static REAL FN(planar_inter_exact)(const FN(pbox)* b1, const FN(pbox)* b2) {
    REAL mx = (b1->x + b2->x) / R_(2), my = (b1->y + b2->y) / R_(2);
    REAL c1 = FN(rcos)(b1->a), s1 = FN(rsin)(b1->a), c2 = FN(rcos)(b2->a), s2 = FN(rsin)(b2->a);
    static const double fx[4] = {0.5, -0.5, -0.5, 0.5}, fy[4] = {0.5, 0.5, -0.5, -0.5};
    FN(pt) poly[16], tmp[16];
    int n = 4;
    for (int i = 0; i < 4; i++) {
        REAL lx = R_(fx[i]) * b1->w, ly = R_(fy[i]) * b1->h;
        poly[i].x = (b1->x - mx) + lx * c1 - ly * s1;
        poly[i].y = (b1->y - my) + lx * s1 + ly * c1;
    }
    /* half-planes of rect 2: n_k . (p - c2) <= h_k */
    REAL nx[4] = {c2, -c2, -s2, s2}, ny[4] = {s2, -s2, c2, -c2};
    REAL hh[4] = {b2->w / R_(2), b2->w / R_(2), b2->h / R_(2), b2->h / R_(2)};
    REAL cx = b2->x - mx, cy = b2->y - my;
    for (int k = 0; k < 4 && n > 0; k++) {
        int m = 0;
        for (int i = 0; i < n; i++) {
            FN(pt) P = poly[i], Q = poly[(i + 1) % n];
            REAL dp = nx[k] * (P.x - cx) + ny[k] * (P.y - cy) - hh[k];
            REAL dq = nx[k] * (Q.x - cx) + ny[k] * (Q.y - cy) - hh[k];
            if (dp <= 0) tmp[m++] = P;
            if ((dp < 0 && dq > 0) || (dp > 0 && dq < 0)) {
                REAL t = dp / (dp - dq);
                tmp[m].x = P.x + t * (Q.x - P.x);
                tmp[m].y = P.y + t * (Q.y - P.y);
                m++;
            }
        }
        n = m;
        for (int i = 0; i < n; i++) poly[i] = tmp[i];
    }
    if (n < 3) return 0;
    REAL area = 0;
    for (int i = 1; i < n - 1; i++)
        area += FN(cross2)(FN(pt_sub)(poly[i], poly[0]), FN(pt_sub)(poly[i + 1], poly[0]));
    area = area < 0 ? -area : area;
    return area / R_(2);
}
static REAL FN(planar_iou_exact)(const FN(pbox)* b1, const FN(pbox)* b2, int mode) {
    REAL inter = FN(planar_inter_exact)(b1, b2);
    REAL a1 = b1->w * b1->h, a2 = b2->w * b2->h;
    return mode == SPH2POB_MODE_IOU ? inter / (a1 + a2 - inter) : inter / a1;
}

static REAL FN(planar_iou)(const FN(pbox)* b1, const FN(pbox)* b2, int mode, int planar) {
    if (planar == SPH2POB_PLANAR_EXACT) return FN(planar_iou_exact)(b1, b2, mode);
    return planar == SPH2POB_PLANAR_MMCV ? FN(planar_iou_mmcv)(b1, b2, mode) : FN(planar_iou_diff)(b1, b2, mode);
}

static inline REAL FN(rmin)(REAL a, REAL b) { return a < b ? a : b; }
static inline REAL FN(rmax)(REAL a, REAL b) { return a > b ? a : b; }

/* Sph-IoU / FoV-IoU closed forms: sphdet/iou/approximate_ious.py:3-54 (inputs after the spherical jitter). */
static REAL FN(approx_iou)(const REAL* g_, const REAL* p_, int fov) {
    REAL g0 = g_[0], p0 = p_[0];
    REAL dt = g0 - p0;
    dt = dt < 0 ? -dt : dt;
    if (dt > R_(180)) { /* standardize_spherical_box :59-79 */
        g0 = (REAL)MFMOD(g0 + R_(180), R_(360));
        p0 = (REAL)MFMOD(p0 + R_(180), R_(360));
    }
    const REAL pi = R_(3.141592653589793), hpi = R_(3.141592653589793 / 2);
    REAL thg = FN(deg2rad)(g0) - pi, phg = hpi - FN(deg2rad)(g_[1]);   /* angle2radian 'convention' :81-99 */
    REAL thp = FN(deg2rad)(p0) - pi, php = hpi - FN(deg2rad)(p_[1]);
    REAL ag = FN(deg2rad)(g_[2]), bg = FN(deg2rad)(g_[3]), ap = FN(deg2rad)(p_[2]), bp = FN(deg2rad)(p_[3]);
    REAL ag2 = ag / R_(2), bg2 = bg / R_(2), ap2 = ap / R_(2), bp2 = bp / R_(2);
    REAL tmin, tmax;
    if (fov) {
        REAL delta = (thp - thg) * FN(rcos)((phg + php) / R_(2));
        tmin = FN(rmax)(-ag2, delta - ap2);
        tmax = FN(rmin)(ag2, delta + ap2);
    } else {
        tmin = FN(rmax)(thg - ag2, thp - ap2);
        tmax = FN(rmin)(thg + ag2, thp + ap2);
    }
    REAL pmin = FN(rmax)(phg - bg2, php - bp2), pmax = FN(rmin)(phg + bg2, php + bp2);
    REAL ai = FN(rmax)(tmax - tmin, 0) * FN(rmax)(pmax - pmin, 0);
    REAL au = ag * bg + ap * bp - ai;
    return ai / (au + R_(1e-8));
}

/* ------------------------------------------------------------------------------------------------ */
/* _sph2pob_iou_auxiliary for ONE pair: sphdet/iou/sph_iou_api.py:48-86 */
static REAL FN(pair_iou)(const REAL* b1_, const REAL* b2_, int dim, int variant, int mode, int edge, int angle,
                         int planar) {
    REAL b1[5], b2[5];
    for (int k = 0; k < dim; k++) { b1[k] = b1_[k]; b2[k] = b2_[k]; }
    FN(jitter_spherical)(b1, b2, dim);
    if (variant == SPH2POB_VARIANT_SPH_IOU || variant == SPH2POB_VARIANT_FOV_IOU) /* sph_iou_api.py:128-175 */
        return FN(clampr)(FN(approx_iou)(b1, b2, variant == SPH2POB_VARIANT_FOV_IOU), 0, 1);
    FN(pbox) p1, p2;
    FN(transform_dispatch)(variant, b1, b2, dim, edge, angle, &p1, &p2);
    FN(jitter_rotated)(&p1, &p2);
    REAL iou = FN(planar_iou)(&p1, &p2, mode, planar);
    return FN(clampr)(iou, 0, 1);
}

/* obb2hbb_xyxy: sphdet/bbox/box_formator.py:33-54 */
static void FN(obb2hbb)(const FN(pbox)* b, REAL out[4]) {
    REAL ca = FN(rcos)(b->a), sa = FN(rsin)(b->a);
    ca = ca < 0 ? -ca : ca; sa = sa < 0 ? -sa : sa;
    REAL W = ca * b->w + sa * b->h, H = sa * b->w + ca * b->h;
    out[0] = b->x - W / R_(2); out[1] = b->y - H / R_(2); out[2] = b->x + W / R_(2); out[3] = b->y + H / R_(2);
}
/* Sph2PobTransfrom.new_forward + obb_iou_loss for ONE pair, unweighted element loss.
 * sphdet/losses/sph2pob_transform.py:24-35 ; sphdet/losses/sph2pob_iou_loss.py:104-196 */
static REAL FN(pair_loss)(const REAL* pred_, const REAL* target_, int dim, int loss_mode, double eps_, REAL* iou_out) {
    REAL b1[5], b2[5];
    for (int k = 0; k < dim; k++) { b1[k] = pred_[k]; b2[k] = target_[k]; }
    FN(jitter_spherical)(b1, b2, dim);
    FN(pbox) p, t;
    FN(transform_standard)(b1, b2, dim, SPH2POB_EDGE_ARC, SPH2POB_ANGLE_EQUATOR, &p, &t);
    FN(jitter_rotated)(&p, &t);
    const REAL eps = R_(eps_);
    REAL ious = FN(clampr)(FN(planar_iou_diff)(&p, &t, SPH2POB_MODE_IOU), 0, 1);
    if (iou_out) *iou_out = ious;
    if (loss_mode == SPH2POB_LOSS_IOU) return R_(1) - FN(clampr)(ious, 0, 1);
    REAL hp[4], ht[4];
    FN(obb2hbb)(&p, hp);
    FN(obb2hbb)(&t, ht);
    REAL ex1 = FN(rmin)(hp[0], ht[0]), ey1 = FN(rmin)(hp[1], ht[1]);
    REAL ex2 = FN(rmax)(hp[2], ht[2]), ey2 = FN(rmax)(hp[3], ht[3]);
    REAL cw = FN(rmax)(ex2 - ex1, 0), ch = FN(rmax)(ey2 - ey1, 0);
    if (loss_mode == SPH2POB_LOSS_GIOU) {
        REAL ix1 = FN(rmax)(hp[0], ht[0]), iy1 = FN(rmax)(hp[1], ht[1]);
        REAL ix2 = FN(rmin)(hp[2], ht[2]), iy2 = FN(rmin)(hp[3], ht[3]);
        REAL iw = FN(rmax)(ix2 - ix1, 0), ih = FN(rmax)(iy2 - iy1, 0);
        REAL ae = cw * ch, ai = iw * ih, ap = p.w * p.h, at = t.w * t.h;
        REAL au = ap + at - ai;
        REAL ratio = (ae - au) / (ae + eps);
        return R_(1) - (ious - FN(clampr)(ratio, 0, 1));
    }
    REAL c2 = cw * cw + ch * ch + eps;
    REAL dx = t.x - p.x, dy = t.y - p.y;
    REAL rho2 = dx * dx + dy * dy;
    if (loss_mode == SPH2POB_LOSS_DIOU) return R_(1) - (ious - FN(clampr)(rho2 / c2, 0, 1));
    const REAL factor = R_(4.0 / (3.141592653589793 * 3.141592653589793));
    REAL dv = (REAL)MATAN(t.w / (t.h + eps)) - (REAL)MATAN(p.w / (p.h + eps));
    REAL v = factor * (dv * dv);
    REAL alpha = (ious > R_(0.5) ? R_(1) : R_(0)) * v / (R_(1) - ious + v + eps);
    return R_(1) - (ious - (FN(clampr)(rho2 / c2, 0, 1) + alpha * v));
}

/* ---------------------------------------- batch entry points ---------------------------------------- */
static int FN(iou_aligned)(const REAL* b1, const REAL* b2, REAL* out, int64_t n, int dim, int variant, int mode,
                           int edge, int angle, int planar, int nthreads) {
    if (dim != 4 && dim != 5) return SPH2POB_ERR_DIM;
    if (variant >= SPH2POB_VARIANT_LEGACY && dim == 5) return SPH2POB_ERR_DIM;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int64_t i = 0; i < n; i++) out[i] = FN(pair_iou)(b1 + i * dim, b2 + i * dim, dim, variant, mode, edge, angle, planar);
    return 0;
}
static int FN(iou_pairwise)(const REAL* b1, int64_t m, const REAL* b2, int64_t n, REAL* out, int dim, int variant,
                            int mode, int edge, int angle, int planar, int nthreads) {
    if (dim != 4 && dim != 5) return SPH2POB_ERR_DIM;
    if (variant >= SPH2POB_VARIANT_LEGACY && dim == 5) return SPH2POB_ERR_DIM;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int64_t k = 0; k < m * n; k++) {
        int64_t i = k / n, j = k % n;
        out[k] = FN(pair_iou)(b1 + i * dim, b2 + j * dim, dim, variant, mode, edge, angle, planar);
    }
    return 0;
}
/* stage outputs for debugging / stage-wise parity: planar boxes after (optional) jitters */
static int FN(transform_batch)(const REAL* b1, const REAL* b2, REAL* o1, REAL* o2, int64_t n, int dim, int variant,
                               int edge, int angle, int jitter) {
    if (dim != 4 && dim != 5) return SPH2POB_ERR_DIM;
    if (variant == SPH2POB_VARIANT_LEGACY && dim == 5) return SPH2POB_ERR_DIM;
    for (int64_t i = 0; i < n; i++) {
        REAL a[5], b[5];
        for (int k = 0; k < dim; k++) { a[k] = b1[i * dim + k]; b[k] = b2[i * dim + k]; }
        if (jitter) FN(jitter_spherical)(a, b, dim);
        FN(pbox) p1, p2;
        FN(transform_dispatch)(variant, a, b, dim, edge, angle, &p1, &p2);
        if (jitter) FN(jitter_rotated)(&p1, &p2);
        memcpy(o1 + i * 5, &p1, sizeof(p1));
        memcpy(o2 + i * 5, &p2, sizeof(p2));
    }
    return 0;
}
static int FN(planar_batch)(const REAL* p1, const REAL* p2, REAL* out, int64_t n, int mode, int planar) {
    for (int64_t i = 0; i < n; i++) {
        FN(pbox) a, b;
        memcpy(&a, p1 + i * 5, sizeof(a));
        memcpy(&b, p2 + i * 5, sizeof(b));
        out[i] = FN(planar_iou)(&a, &b, mode, planar);
    }
    return 0;
}
static int FN(loss_batch)(const REAL* pred, const REAL* target, REAL* loss, REAL* iou, int64_t n, int dim,
                          int loss_mode, double eps, int nthreads) {
    if (dim != 4 && dim != 5) return SPH2POB_ERR_DIM;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int64_t i = 0; i < n; i++)
        loss[i] = FN(pair_loss)(pred + i * dim, target + i * dim, dim, loss_mode, eps, iou ? iou + i : NULL);
    return 0;
}

#undef R_

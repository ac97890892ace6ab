// Unbiased IoU of two spherical rectangles (exact area of the intersection polygon on the unit sphere) for gfx950.
//
// Replaces sphdet/iou/sph_iou_api.py:103-126 (`unbiased_iou`) over sphdet/iou/unbiased_iou_bfov.py:4-204 (BFoV) and
// sphdet/iou/unbiased_iou_rbfov.py:4-182 (RBFoV, incl. roll_T :10-34), which run as numpy on the CPU in the reference
// (README: 46 s per 1 M pairs).  One lane owns one pair, double precision throughout (the reference's own comment,
// unbiased_iou_bfov.py:187: "This program need high-precision float operator!!"):
//
//   1. each box -> 4 great-circle plane normals [left, right, up, down] (getNormal :13-47), RBFoV: rotated about the
//      look-at axis by gamma (roll_T);
//   2. candidate vertices of the intersection: the 8 box corners and, for each of the 16 (plane of box 1, plane of
//      box 2) pairs, the two antipodal intersection points of the great circles (remove_outer_points :105-139);
//   3. a candidate is kept when np.round(P . N_k, 8) >= 0 for all 8 normals; each kept vertex contributes the angle
//      acos(-N_a . N_b) between the two planes that meet there; area = sum - (count - 2) * pi (interArea :49-62).
//
// Identities used (exact in real arithmetic, 1e-16-level in floating point; never at a decision threshold except
// where the reference's own decision is already rounding noise):
//   * the two antipodal points share their 8 dot products up to sign: one pass gives both membership tests;
//   * np.round(d / (|t| + 1e-10), 8) >= 0  <=>  d * 1e8 >= -0.5 * (|t| + 1e-10): no normalisation, no divisions;
//   * a vertex lies on its own two planes: those two dot products are not evaluated.
//
// REFPREC = false (default arithmetic): fp32 spherical jitter and fp32 deg2rad exactly as the reference applies them
//   to fp32 tensors (sph_iou_api.py:121, unbiased_iou_bfov.py:189), everything after that in double.
// REFPREC = true (SPH2POB_FLAG_REFERENCE_ORDER): additionally rounds to fp32 wherever numpy keeps float32 for the
//   reference's float32 inputs (sin/cos of the inputs, V_lookat, V_up, N_up, N_down, rotation entries, both areas).
#pragma once
#include "sph2pob_device.hpp"

namespace sph2pob {

struct DVec {
    double x, y, z;
};
SPH_DEV DVec dcross(const DVec& a, const DVec& b) {
    return DVec{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
SPH_DEV double ddot(const DVec& a, const DVec& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

template <bool R>
SPH_DEV double ubf(double x) {
    return R ? (double)(float)x : x;
}
// sin and cos of |x| <~ 8 rad in double to ~1 ulp: quadrant reduction by pi/2 in two pieces (the first has 33
// significant bits, so k * PIO2_1 is exact for the |k| <= 5 that occur here) and the fdlibm kernel polynomials on
// [-pi/4, pi/4].  The library sincos carries a Payne-Hanek path for huge arguments and costs ~195 instructions, ten
// calls per pair: a third of the kernel.
SPH_DEV void sincos_d(double x, double& s, double& c) {
    const double k = rint(x * 6.36619772367581382433e-01);             // 2 / pi
    double r = fma(-k, 1.57079632673412561417e+00, x);                 // PIO2_1
    r = fma(-k, 6.07710050650619224932e-11, r);                        // PIO2_1T
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double sp = fma(r * z, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double cp = w + (((1.0 - w) - hz) + z * (z * pc));
    const int q = (int)k & 3;
    const double a = (q & 1) ? cp : sp, b = (q & 1) ? sp : cp;
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}

template <bool R>
SPH_DEV void ub_sincos(double x, double& s, double& c) {
    if (R) {
        s = (double)sinf((float)x);
        c = (double)cosf((float)x);
    } else {
        sincos_d(x, s, c);
    }
}
SPH_DEV double ub_acos_clip(double c) { return acos(fmin(fmax(c, -1.0), 1.0)); }

struct UbBox {
    DVec n[4];       // left, right, up, down
    DVec look;
    double corner;   // interior angle at each corner
    double area;
    double cos_r;    // cosine of the circumradius (corner distance from the centre), < 0: unknown / do not cull
    double sin_r;
};

// roll_T (unbiased_iou_rbfov.py:10-34).  LO: the operand is a float32 array in the reference (N_up / N_down).
template <bool R, bool LO>
SPH_DEV DVec ub_roll(const double (&m)[3][3], const DVec& p) {
    constexpr bool Q = R && LO;
    DVec r;
    r.x = ubf<Q>(ubf<Q>(ubf<Q>(m[0][0] * p.x) + ubf<Q>(m[0][1] * p.y)) + ubf<Q>(m[0][2] * p.z));
    r.y = ubf<Q>(ubf<Q>(ubf<Q>(m[1][0] * p.x) + ubf<Q>(m[1][1] * p.y)) + ubf<Q>(m[1][2] * p.z));
    r.z = ubf<Q>(ubf<Q>(ubf<Q>(m[2][0] * p.x) + ubf<Q>(m[2][1] * p.y)) + ubf<Q>(m[2][2] * p.z));
    return r;
}

// rad: (theta, phi, alpha, beta[, gamma]) in radians, already rounded to fp32 as the reference hands them to numpy
template <int DIM, bool R>
SPH_DEV void ub_box(const float (&rad)[5], UbBox& B) {
    const double th = rad[0], ph = rad[1];
    const double a2 = ubf<R>((double)rad[2] / 2), b2 = ubf<R>((double)rad[3] / 2);
    double st, ct, sp, cp, sa, ca, sb, cb;
    ub_sincos<R>(th, st, ct);
    ub_sincos<R>(ph, sp, cp);
    ub_sincos<R>(a2, sa, ca);
    ub_sincos<R>(b2, sb, cb);
    const DVec look{ubf<R>(sp * ct), ubf<R>(sp * st), cp};
    const DVec right{-st, ct, 0.0};
    const DVec up{ubf<R>(-cp * ct), ubf<R>(-cp * st), sp};
    B.look = look;
    // left / right: float32 scalar * float64 V_right stays float64, the sin * V_lookat product is float32 (:28-29)
    B.n[0] = DVec{-ca * right.x + ubf<R>(sa * look.x), -ca * right.y + ubf<R>(sa * look.y), -ca * right.z + ubf<R>(sa * look.z)};
    B.n[1] = DVec{ca * right.x + ubf<R>(sa * look.x), ca * right.y + ubf<R>(sa * look.y), ca * right.z + ubf<R>(sa * look.z)};
    // up / down: float32 throughout (:30-31)
    B.n[2] = DVec{ubf<R>(ubf<R>(-cb * up.x) + ubf<R>(sb * look.x)), ubf<R>(ubf<R>(-cb * up.y) + ubf<R>(sb * look.y)),
                  ubf<R>(ubf<R>(-cb * up.z) + ubf<R>(sb * look.z))};
    B.n[3] = DVec{ubf<R>(ubf<R>(cb * up.x) + ubf<R>(sb * look.x)), ubf<R>(ubf<R>(cb * up.y) + ubf<R>(sb * look.y)),
                  ubf<R>(ubf<R>(cb * up.z) + ubf<R>(sb * look.z))};
    if (DIM == 5) {
        double sg, cg;
        ub_sincos<R>((double)rad[4], sg, cg);
        const double omc = ubf<R>(1.0 - cg);
        const double nx = look.x, ny = look.y, nz = look.z;
        double m[3][3];
        m[0][0] = ubf<R>(ubf<R>(ubf<R>(nx * nx) * omc) + cg);
        m[0][1] = ubf<R>(ubf<R>(ubf<R>(nx * ny) * omc) - ubf<R>(nz * sg));
        m[0][2] = ubf<R>(ubf<R>(ubf<R>(nx * nz) * omc) + ubf<R>(ny * sg));
        m[1][0] = ubf<R>(ubf<R>(ubf<R>(nx * ny) * omc) + ubf<R>(nz * sg));
        m[1][1] = ubf<R>(ubf<R>(ubf<R>(ny * ny) * omc) + cg);
        m[1][2] = ubf<R>(ubf<R>(ubf<R>(ny * nz) * omc) - ubf<R>(nx * sg));
        m[2][0] = ubf<R>(ubf<R>(ubf<R>(nx * nz) * omc) - ubf<R>(ny * sg));
        m[2][1] = ubf<R>(ubf<R>(ubf<R>(ny * nz) * omc) + ubf<R>(nx * sg));
        m[2][2] = ubf<R>(ubf<R>(ubf<R>(nz * nz) * omc) + cg);
        B.n[0] = ub_roll<R, false>(m, B.n[0]);
        B.n[1] = ub_roll<R, false>(m, B.n[1]);
        B.n[2] = ub_roll<R, true>(m, B.n[2]);
        B.n[3] = ub_roll<R, true>(m, B.n[3]);
    }
    if (R) {  // area :10-12 in float32; corner angle from the (partly float32) normals as interArea does
        B.corner = ub_acos_clip(-ddot(B.n[0], B.n[2]));
        const float s = -sinf(rad[2] / 2) * sinf(rad[3] / 2);
        B.area = (double)(4 * acosf(s) - (float)(2 * 3.141592653589793));
    } else {
        // N_left . N_up = sin(a/2) sin(b/2) (the rotation preserves it): one acos serves the four corner angles and
        // the area 4 acos(-sin sin) - 2 pi; the dot product of the rounded normals differs from it by ~1e-16
        B.corner = acos(-sa * sb);
        B.area = 4.0 * B.corner - 2.0 * 3.141592653589793;
    }
    // circumradius for the disjointness cull: corners sit at (+-tan a2, +-tan b2, 1) in the box frame
    if (ca > 0.02 && cb > 0.02) {
        const double ta = sa / ca, tb = sb / cb;
        B.cos_r = 1.0 / sqrt(1.0 + ta * ta + tb * tb);
        B.sin_r = sqrt(fmax(1.0 - B.cos_r * B.cos_r, 0.0));
    } else {
        B.cos_r = -2.0;
        B.sin_r = 0.0;
    }
}

// membership of the candidate direction t (not normalised, |t| + 1e-10 = nrm) and of its antipode, against the
// normals listed; lo/hi accumulate min / max of t . N
SPH_DEV void ub_minmax(const DVec& t, const DVec& n, double& lo, double& hi) {
    const double d = ddot(t, n);
    lo = fmin(lo, d);   // fmin/fmax drop NaN: handled by the caller through nrm
    hi = fmax(hi, d);
}

// np.round(d / (|t| + off), 8) >= 0 for the smallest (most negative) dot product d = lo <= 0 of a candidate, i.e.
// lo * 1e8 >= -0.5 * (sqrt(tt) + off).  |t| <= ~1 (cross product of unit normals), so the square root is only needed
// inside the 5e-9 band where the reference's rounding decides; NaN (tt != tt) is never inside.
SPH_DEV bool ub_inside(double lo, double tt, double off) {
    if (!(tt == tt)) return false;
    if (lo >= 0.0) return true;
    if (lo * 1e8 < -0.51) return false;
    return lo * 1e8 >= -0.5 * (sqrt(tt) + off);
}

template <int DIM, bool R>
SPH_DEV float unbiased_pair_iou(const float (&in1)[5], const float (&in2)[5]) {
    float j1[5], j2[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { j1[k] = in1[k]; j2[k] = in2[k]; }
    jitter_spherical<DIM>(j1, j2);
#pragma unroll
    for (int k = 0; k < 5; k++) { j1[k] = j1[k] * kDeg2Rad; j2[k] = j2[k] * kDeg2Rad; }
    UbBox A, B;
    ub_box<DIM, R>(j1, A);
    ub_box<DIM, R>(j2, B);

    int count = 0;
    double sum = 0.0;
    // boxes whose circumscribed caps are disjoint share no vertex: the candidate loop would keep nothing
    const bool far = A.cos_r > -1.0 && B.cos_r > -1.0 &&
                     ddot(A.look, B.look) < A.cos_r * B.cos_r - A.sin_r * B.sin_r - 1e-6;
    if (!far) {
        // corners of A against B's planes and corners of B against A's (own planes: on them by construction; the
        // opposite own planes are evaluated so that degenerate boxes behave as in the reference)
#pragma unroll
        for (int side = 0; side < 2; side++) {
            const UbBox& P = side == 0 ? A : B;
            const UbBox& Q = side == 0 ? B : A;
            constexpr int ea[4] = {0, 3, 2, 1}, eb[4] = {2, 0, 1, 3};  // [left,up] [down,left] [up,right] [right,down]
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const DVec t = dcross(P.n[ea[c]], P.n[eb[c]]);
                const double tt = ddot(t, t);                  // corners: no 1e-10 (getNormal :41-43)
                double lo = 0.0, hi = 0.0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (k != ea[c] && k != eb[c]) ub_minmax(t, P.n[k], lo, hi);
                    ub_minmax(t, Q.n[k], lo, hi);
                }
                if (tt > 0.0 && ub_inside(lo, tt, 0.0)) {
                    sum += P.corner;
                    count++;
                }
            }
        }
        // the 16 plane pairs: t = N_i x N'_j, candidates +t and -t
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const DVec t = dcross(A.n[i], B.n[j]);
                const double tt = ddot(t, t);
                double lo = 0.0, hi = 0.0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (k != i) ub_minmax(t, A.n[k], lo, hi);
                    if (k != j) ub_minmax(t, B.n[k], lo, hi);
                }
                const bool in_p = ub_inside(lo, tt, 1e-10), in_q = ub_inside(-hi, tt, 1e-10);
                if (in_p || in_q) {
                    const double ang = ub_acos_clip(-ddot(A.n[i], B.n[j]));
                    const int c = (in_p ? 1 : 0) + (in_q ? 1 : 0);
                    sum += ang * c;
                    count += c;
                }
            }
        }
    }
    const double inter = count ? sum - (double)(count - 2) * 3.141592653589793 : 0.0;
    const double au = R ? (double)((float)A.area + (float)B.area) : A.area + B.area;
    const double eps = 1e-8;
    const double iou = DIM == 4 ? (inter + eps) / (au - (inter + eps))   // unbiased_iou_bfov.py:200
                                : inter / (au - inter + eps);             // unbiased_iou_rbfov.py:178
    const float f = (float)iou;                                           // .float(); clamp(0, 1) sph_iou_api.py:126
    return f != f ? f : fminf(fmaxf(f, 0.0f), 1.0f);
}

}  // namespace sph2pob

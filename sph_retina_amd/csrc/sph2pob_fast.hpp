// Fast geometry core for the default Sph2Pob IoU path (variants standard / efficient, rbb_angle='equator').
//
// sph2pob_standard and sph2pob_efficient produce the SAME pair of planar boxes up to a rigid translation
// (g at (-A/2, pi/2) | (0, 0), p at (+A/2, pi/2) | (A, 0), same widths, same angles), so one
// core serves both.  Instead of re-tracing the reference's ~100 tensor ops (cross products, six F.normalize =
// 18 IEEE divides, 3-6 acos, 4 planar sin/cos), the core evaluates the same quantities in closed form:
//
//     N  =  c_p . d_g ,  D  = -c_p . e_g        (d = meridian tangent, e = c x d = east; bearing of p seen from g)
//     N' = -c_g . d_p ,  D' =  c_g . e_p
//     sin A = |(N, D)| ,  cos A = c_g . c_p      A = great-circle distance of the centres
//     (cos a_g, sin a_g) = (D , N ) / sin A      — the reference's  sign * acos(clamp(d^ . z^))  (sph2pob_efficient.py:81-97)
//     (cos a_p, sin a_p) = (D', N') / sin A
//
// with the theta difference taken BEFORE the trig (sin/cos of (theta_p - theta_g)/2, haversine style), which is
// better conditioned than the reference's products of separately rounded sin/cos — the core sits below the
// reference's own fp32 noise floor.  The planar angles are never formed: the clipping stage needs
// only (cos, sin); gamma (RBFoV) and the jitter's constant angle bumps are applied as rotations of (cos, sin).
// What IS mirrored exactly: both jitters (decisions on the same quantities with the same thresholds), the rounding
// of deg->rad, and the acos(clamp(., +-(1 - 1e-7))) floors of the reference (A >= 4.88e-4 [efficient] /
// 9.77e-4 [standard], |sin a| >= 4.88e-4).  Pairs whose bounding circles cannot touch return 0 early — exact,
// because disjoint rectangles give exactly 0 in the reference.
#pragma once
#include "sph2pob_device.hpp"

namespace sph2pob {

SPH_DEV unsigned float_bits(float x) { return __builtin_bit_cast(unsigned, x); }
SPH_DEV unsigned max3_u32(unsigned a, unsigned b, unsigned c) {
    const unsigned m = a > b ? a : b;
    return m > c ? m : c;   // one v_max3_u32
}
SPH_DEV float fast_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
SPH_DEV float fast_rsq(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(x);
#else
    return 1.0f / sqrtf(x);
#endif
}

// hardware sin/cos (v_sin_f32 / v_cos_f32, argument in revolutions, |abs err| of a few 1e-6): ONLY used by the
// conservative cull test, never for a value that reaches the output
SPH_DEV float hw_sin_rev(float rev) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sinf(rev);
#else
    return sinf(rev * 6.283185307179586f);
#endif
}
SPH_DEV float hw_cos_rev(float rev) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_cosf(rev);
#else
    return cosf(rev * 6.283185307179586f);
#endif
}

// sin and cos of x (|x| <= ~8 rad) to ~1 ulp: Cody-Waite reduction by pi/2 + Cephes minimax polynomials.
SPH_DEV void sincos_r(float x, float& s, float& c) {
    float k = rintf(x * 0.63661977236758134f);
    float r = fmaf(-k, 1.5703125f, x);
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.549789954891882e-8f, r);
    float z = r * r;
    float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float sp = fmaf(r * z, ps, r);
    float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float cp = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
#if defined(__HIP_DEVICE_COMPILE__)
    int q = (int)k & 3;                         // v_cvt_i32_f32 saturates (NaN -> 0): defined on the device
#else
    int q = (k == k ? (int)k : 0) & 3;          // the host build (unit tests, UBSan): a NaN angle must not reach the cast
#endif
    float a = (q & 1) ? cp : sp;
    float b = (q & 1) ? sp : cp;
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}

// The same reduction and polynomials with cheaper quadrant logic for the two argument ranges of stage 1 (the generic
// selection above costs 12 compare / select / integer instructions per call, as much as the polynomials themselves):
// sin / cos of a colatitude x in [0, pi] — what the spherical jitter's clamps guarantee: k in {0, 1, 2}, sin >= 0.
// Bit-identical to sincos_r on that range.
SPH_DEV void sincos_colat(float x, float& s, float& c) {
    float k = rintf(x * 0.63661977236758134f);
    float r = fmaf(-k, 1.5703125f, x);
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.549789954891882e-8f, r);
    float z = r * r;
    float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float sp = fmaf(r * z, ps, r);
    float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float cp = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
    const bool odd = k == 1.0f;
    const float a = odd ? cp : sp, b = odd ? sp : cp;
    s = fabsf(a);               // k = 2: r in [-pi/4, 0], sin x = -sin r >= 0
    c = k >= 1.0f ? -b : b;     // cos(pi/2 + r) = -sin r, cos(pi + r) = -cos r
}
// sin(2x) and 1 - cos(2x) for x in [-pi, pi] (half the longitude difference of two clamped longitudes): both are
// invariant under (sin x, cos x) -> (-sin x, -cos x), so only the parity of the quadrant matters.  Bit-identical to
// 2 sh ch / 2 sh sh from sincos_r on that range.
SPH_DEV void sin_vers_double(float x, float& sin2x, float& vers2x) {
    float k = rintf(x * 0.63661977236758134f);
    float r = fmaf(-k, 1.5703125f, x);
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.549789954891882e-8f, r);
    float z = r * r;
    float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float sp = fmaf(r * z, ps, r);
    float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float cp = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
    const bool odd = fabsf(k) == 1.0f;
    const float t = sp * cp, t2 = t + t;
    sin2x = odd ? -t2 : t2;
    const float a = odd ? cp : sp;
    vers2x = 2.0f * a * a;
}

// atan(t), 0 <= t <= 1 (|err| <= 1.3e-7): odd polynomial of degree 17, near-minimax fit.
SPH_DEV float atan_unit(float t) {
    float u = t * t;
    float p = 0.0028340641874819994f;
    p = fmaf(p, u, -0.016005029901862144f);
    p = fmaf(p, u, 0.042587608098983765f);
    p = fmaf(p, u, -0.07495445758104324f);
    p = fmaf(p, u, 0.10636754333972931f);
    p = fmaf(p, u, -0.14202570915222168f);
    p = fmaf(p, u, 0.19992484152317047f);
    p = fmaf(p, u, -0.3333306610584259f);
    return fmaf(p * u, t, t);
}
SPH_DEV float atan2_r(float y, float x) {
    float ax = fabsf(x), ay = fabsf(y);
    float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float t = mx > 0.0f ? mn * fast_rcp(mx) : 0.0f;
    t = fminf(t, 1.0f);
    float p = atan_unit(t);
    p = ay > ax ? 1.57079632679489662f - p : p;
    p = x < 0.0f ? 3.14159265358979324f - p : p;
    return y < 0.0f ? -p : p;
}

// atan2(y, x) for y >= 0 and x^2 + y^2 ~ 1 (the angle between two unit vectors from |cross| and dot): the same
// quotient, polynomial and octant fix-ups as atan2_r without the selects that cannot fire on that domain
// (the 0 / 0 guard, the clamp of the quotient at 1 — min / max <= 1 up to one rounding of the product — and y < 0)
SPH_DEV float atan2_pos(float y, float x) {
    const float ax = fabsf(x);
    const float mx = fmaxf(ax, y), mn = fminf(ax, y);
    const float t = mn * fast_rcp(mx);
    float p = atan_unit(t);
    p = y > ax ? 1.57079632679489662f - p : p;
    return x < 0.0f ? 3.14159265358979324f - p : p;
}

constexpr float kMinAng = 4.8828125e-4f;     // acos(float(1 - 1e-7))
constexpr float kCosMinAng = 0.99999988f;    // cos(kMinAng) in fp32 = the clamp bound itself

// rotate (c, s) by +ang where (ca, sa) = (cos ang, sin ang)
// (explicit FMAs, contraction off: lean_finish is inlined several times per kernel — in-loop and tail copies, aligned /
// pairwise / NMS kernels — and a pair must get the same bits whichever copy finishes it; left to the compiler, the
// choice of which product of a*b + c*d is fused differs between copies)
SPH_DEV void rot(float& c, float& s, float ca, float sa) {
    const float c2 = fmaf(c, ca, -(s * sa)), s2 = fmaf(s, ca, c * sa);
    c = c2;
    s = s2;
}

// the reference's  sign * |acos(clamp(cos a))|  expressed on (cos a, sin a): floor |a| and |pi - a| at kMinAng;
// sin a == 0 takes the negative sign (criterion `< 0` false -> -1, sph2pob_efficient.py:211-226)
SPH_DEV void angle_floor(float& c, float& s) {
    if (fabsf(s) < kMinAng) {
        s = s > 0.0f ? kMinAng : -kMinAng;
        c = c < 0.0f ? -kCosMinAng : kCosMinAng;
    }
}

// Clipped length of one rectangle edge inside the other rectangle's slabs, pre-sorted-bound form: the edge is
// P(tau) = k + tau * u^ (tau in [0, len]); per axis tau lies in [m - r, m + r] with m = -k_axis / u^_axis and
// r = h_axis * |1 / u^_axis|.  Reciprocals are clamped to +-1e18 so that parallel edges give finite, correctly
// ordered bounds (inside: (-huge, +huge); outside: both bounds on one side => empty) without NaNs.
SPH_DEV float clip_len3(float mx, float my, float rx, float ry, float len) {
    float lo = fmaxf(fmaxf(mx - rx, my - ry), 0.0f);
    float hi = fminf(fminf(mx + rx, my + ry), len);
    return fmaxf(hi - lo, 0.0f);
}
// Twice the boundary-integral contribution of rectangle X's four edges clipped to rectangle Y (Y's frame):
// (px, py) = X's centre, half extents hw/hh (full w/h), axes rotated by (c, s); Y = [-hwy, hwy] x [-hhy, hhy];
// ic = 1/c, is = 1/s (clamped), aic/ais their magnitudes.  Corners are formed first and only then scaled by the
// reciprocals (m = -k_axis / u^_axis): expanding k * (1/u) into sums of large terms would lose the small
// differences that decide near-parallel, near-coincident edges (identical boxes after the jitter).
SPH_DEV float edges_inside3(float px, float py, float c, float s, float ic, float is, float aic, float ais, float hw,
                            float hh, float hwy, float hhy, float w, float h, bool with_origin_terms) {
    // explicit FMAs, contraction off (see rot)
    const float ux = hw * c, uy = hw * s, vx = -hh * s, vy = hh * c;
    const float k0x = (px + ux) + vx, k0y = (py + uy) + vy;
    const float k1x = (px - ux) + vx, k1y = (py - uy) + vy;
    const float k2x = fmaf(2.0f, px, -k0x), k2y = fmaf(2.0f, py, -k0y);
    const float k3x = fmaf(2.0f, px, -k1x), k3y = fmaf(2.0f, py, -k1y);
    const float rux = hwy * aic, ruy = hhy * ais, rvx = hwy * ais, rvy = hhy * aic;
    // e0: k0, dir -u^ = (-c, -s); e1: k1, dir -v^ = (s, -c); e2: k2, dir +u^; e3: k3, dir +v^ = (-s, c)
    const float l0 = clip_len3(k0x * ic, k0y * is, rux, ruy, w);
    const float l1 = clip_len3(-k1x * is, k1y * ic, rvx, rvy, h);
    const float l2 = clip_len3(-k2x * ic, -k2y * is, rux, ruy, w);
    const float l3 = clip_len3(k3x * is, -k3y * ic, rvx, rvy, h);
    if (!with_origin_terms) return fmaf(hh, l0 + l2, hw * (l1 + l3));
    const float xu = fmaf(px, s, -(py * c)), xv = fmaf(px, c, py * s);
    return fmaf(l0, hh - xu, l2 * (hh + xu)) + fmaf(l1, hw - xv, l3 * (hw + xv));
}

// Intersection area of two NEARLY PARALLEL (or nearly perpendicular) rectangles, fp32, first order in the small
// angle: B = [-X, X] x [-Y, Y]; A has centre (px, py), half extents (hw, hh) and axis (c, s) in B's frame with
// |s| < kNearParallel or |c| < kNearParallel.  The reference's two jitter steps can cancel and leave the planar boxes
// parallel to ~1e-6 rad; the boundary integral then evaluates each crossing of nearly coincident edges twice and the
// two fp32 results disagree by 1e-7 / |s| along the edge (up to 2e-2 IoU).  Here A's four edges are straight lines
// y = const + t x, x = const - t y with slope |t| < 2.5e-4, and
//     area = integral over x of [min(Y, top(x)) - max(-Y, bottom(x))]  over the zeroth-order x-range
//            - integral over y of what max(-X, left(y)) and min(X, right(y)) cut off that range,
// every term well conditioned (no division by the angle); the neglected corner terms are O(t^2 L^2) ~ 1e-7 of the area.
SPH_DEV float lin_excess(float fl, float fr, float lim, float W) {
    // integral over a width W of max(f - lim, 0) for f linear from fl to fr
    const float fmin = fminf(fl, fr), fmax = fmaxf(fl, fr), span = fmax - fmin;
    const float e = fminf(fmaxf(fmax - lim, 0.0f), span);
    const float part = span > 0.0f ? 0.5f * e * e * W / span : 0.0f;
    return lim <= fmin ? (0.5f * (fmin + fmax) - lim) * W : part;
}
// A CALLED function (values in, one value out; `lin_excess` inlined into it): inlined, its ~250 instructions and their registers weighed
// on the common path of every finishing pass although one pass in thirty enters it — 7.94 -> 7.75 us per 1 M pairs,
// 50.9 -> 49.7 us per 8 M, the same bits (profiles/r03a_ab_nearcall_*.log)
#if defined(SPH_NEAR_INLINE)
SPH_DEV
#else
__host__ __device__ __attribute__((noinline)) inline
#endif
float near_parallel_inter(float px, float py, float c, float s, float hw, float hh, float X,
                                                                               float Y) {
    float a = hw, b = hh;
    if (fabsf(c) < fabsf(s)) {   // nearly perpendicular: same rectangle, axes rotated by a quarter turn, extents swapped
        const float c2 = s > 0.0f ? s : -s, s2 = s > 0.0f ? -c : c;
        c = c2; s = s2; a = hh; b = hw;
    } else if (c < 0.0f) {
        c = -c; s = -s;
    }
    const float ic = 1.0f / c, t = s * ic;
    // A's four edges as straight lines: top / bottom y = tp0 | bt0 + (x - px) t, left / right x = l0 | r0 - (y - py) t
    const float l0 = px - a * ic, r0 = px + a * ic, bt0 = py - b * ic, tp0 = py + b * ic;
    // zeroth-order overlap box (t = 0): its ranges are where the first-order terms are integrated
    const float xl = fmaxf(-X, l0), xr = fminf(X, r0), yl = fmaxf(-Y, bt0), yr = fminf(Y, tp0);
    const float W = xr - xl, H = yr - yl;
    if (!(W > 0.0f) || !(H > 0.0f)) return 0.0f;
    const float tl = tp0 + (xl - px) * t, tr = tp0 + (xr - px) * t, bl = tl - 2.0f * b * ic, br = tr - 2.0f * b * ic;
    const float top = 0.5f * (tl + tr) * W - lin_excess(tl, tr, Y, W);      // integral over x of min(Y, top(x))
    const float bot = 0.5f * (bl + br) * W + lin_excess(-bl, -br, Y, W);    // integral over x of max(-Y, bottom(x))
    const float ll = l0 - (yl - py) * t, lr = l0 - (yr - py) * t, rl = ll + 2.0f * a * ic, rr = lr + 2.0f * a * ic;
    const float lft = 0.5f * (ll + lr) * H + lin_excess(-ll, -lr, X, H);    // integral over y of max(-X, left(y))
    const float rgt = 0.5f * (rl + rr) * H - lin_excess(rl, rr, X, H);      // integral over y of min(X, right(y))
    // area of the box, with its horizontal sides replaced by the true ones, minus what the true vertical sides remove
    return fmaxf((top - bot) - (lft - xl * H) - (xr * H - rgt), 0.0f);
}

// Stage 0 (cull): a conservative bounding-circle test with hardware trig on the raw boxes.  Returns true when the
// pair's IoU is exactly 0 (the two planar rectangles' circumscribed circles cannot touch, whatever the rounding of
// the accurate path: the bound carries 1.5e-3 rad for both jitters + the reference's own rounding of A, and 1e-4 in
// cos-space for the hardware trig's error).
// Per-box part of the cull: hardware sin/cos of the colatitude, longitude in revolutions, circumscribed-circle
// radius of the planar rectangle (+inf for a degenerate box so that it is never culled).
struct CullBox { float s, c, th_rev, r; };
SPH_DEV CullBox cull_box(const float (&b)[5], int edge) {
#pragma clang fp contract(fast)
    // raw (un-jittered) box: the spherical jitter moves every coordinate by at most 2.5e-4 deg (4.3e-6 rad) and the
    // clamps only shrink extents; both are far inside the 1.5e-3 rad margin.  Extents are clamped to the jitter's
    // upper bound (180 deg) so that an out-of-range alpha/beta cannot under-estimate the radius; phi is clamped
    // into [0, 180] like the jitter does; theta enters only through cos(theta_p - theta_g) (periodic).
    // (NaN-propagating clamps: a NaN coordinate makes the pair's test false, i.e. the pair is never culled)
    float w = min_nan(b[2], 180.0f) * kDeg2Rad, h = min_nan(b[3], 180.0f) * kDeg2Rad;
    if (edge != EDGE_ARC) { w = edge_length(w, edge); h = edge_length(h, edge); }
    float d = w * w + h * h;
    const float kRev = 1.0f / 360.0f;
    float ph = min_nan(max_nan(b[1], 0.0f), 180.0f) * kRev;
    CullBox cb;
    cb.s = hw_sin_rev(ph);
    cb.c = fmaf(b[4], 0.0f, hw_cos_rev(ph));   // a NaN / infinite gamma (0 for BFoV: folded) must not be culled either
    cb.th_rev = min_nan(max_nan(b[0], 0.0f), 360.0f) * kRev;
    cb.r = d > 0.0f ? 0.5f * d * fast_rsq(d) : __builtin_inff();   // d = NaN -> +inf: never culled
    return cb;
}
// Per-pair part: true when the circumscribed circles cannot touch (IoU exactly 0).  1 - R^2/2 + R^4/24 - R^6/720
// <= cos R; margins: 1.5e-3 rad for both jitters + the reference's rounding of A, 1e-4 in cos-space for the
// hardware trig.
SPH_DEV bool cull_pair(const CullBox& g, const CullBox& p) {
#pragma clang fp contract(fast)
    float R = (g.r + p.r) + 1.5e-3f;
    float R2 = R * R;
    float cosR_lb = fmaf(fmaf(fmaf(-1.0f / 720.0f, R2, 1.0f / 24.0f), R2, -0.5f), R2, 1.0f);
    float C = g.c * p.c + (g.s * p.s) * hw_cos_rev(p.th_rev - g.th_rev);
    // (no separate test of R: the bound decreases in R^2 on the whole axis and is below -1.12 from R^2 = 8.9 on, where
    // no cosine can be below it — see fast_cull; R = +inf or NaN makes the bound -inf or NaN: never culled)
    return C < cosR_lb - 1e-4f;
}
// Aligned form of the same test with 4 quarter-rate instructions instead of 7 (a quarter-rate op costs 8 plain FMAs):
//   cos(angular distance) = 1/2 [cos(phi_g - phi_p) (1 + cos dtheta) + cos(phi_g + phi_p) (1 - cos dtheta)]   (3 v_cos)
//   (r_g + r_p)^2 = (d_g + d_p + 2 sqrt(d_g d_p)) / 4,  d = w^2 + h^2                                        (1 v_rsq)
// and the 1.5e-3 rad margin moved into R^2: (R0 + m)^2 <= R0^2 + 3 m + m^2 for R0 < 3 (cos R's polynomial lower bound
// decreases in R^2 on the whole range, so a larger R^2 only culls less).  A degenerate box (d = 0) gives 0 * inf = NaN
// and is never culled, like the +inf radius of cull_box.
// CHORD (sph2pob_legacy): that transform places the planar centres at a distance d with d >= 2 sin(L / 2), the chord of
// the great-circle distance L, not at L itself (d^2 / 4 = t^2 + b^2 >= sin^2 t + sin^2 b >= sin^2(L / 2) with b = half the
// latitude difference and sin t = sqrt(sin^2(L/2) - sin^2 b) / cos b: sph2pob_legacy.py:38-83), so the circles are
// certainly apart only when the chord exceeds the sum of the radii: cos L < 1 - R^2 / 2.
template <int DIM, bool CHORD = false>
SPH_DEV bool fast_cull(const float (&g)[5], const float (&p)[5], int edge) {
#pragma clang fp contract(fast)
    // Every coordinate inside the range the spherical jitter clamps it to (theta in [0, 360], phi and the extents in
    // [0, 180] degrees)?  Tested on the bit patterns as unsigned integers — for non-negative floats the order of the
    // patterns is the order of the values, and a negative number, an infinity or a NaN has a pattern above that of any
    // bound — with 2 v_max3_u32 + 2 v_max_u32 + 2 compares for the eight coordinates.  A pair with a coordinate out of
    // range (or NaN) is never culled: the finishing stage clamps it exactly like the reference does.  (Round 1 clamped
    // all eight values here instead, NaN-propagating: 12 v_minimum3 / v_maximum3.)
    const unsigned mg = max3_u32(float_bits(g[1]), float_bits(g[2]), float_bits(g[3]));
    const unsigned mp = max3_u32(float_bits(p[1]), float_bits(p[2]), float_bits(p[3]));
    const unsigned mth = float_bits(g[0]) > float_bits(p[0]) ? float_bits(g[0]) : float_bits(p[0]);
    const bool in_range = ((mg > mp ? mg : mp) <= 0x43340000u) & (mth <= 0x43b40000u);   // 180.0f, 360.0f
    // (r_g + r_p)^2 with r = half the diagonal: (d_g + d_p + 2 sqrt(d_g d_p)) / 4, d = w^2 + h^2.  For arc edges the
    // degrees -> radians factor is folded into the last FMA; a degenerate box (d = 0) gives 0 * inf = NaN: never culled.
    float dg, dp, quarter;
    if (edge == EDGE_ARC) {
        dg = fmaf(g[2], g[2], g[3] * g[3]);
        dp = fmaf(p[2], p[2], p[3] * p[3]);
        quarter = 0.25f * kDeg2Rad * kDeg2Rad;
    } else {
        const float wg = edge_length(g[2] * kDeg2Rad, edge), hg = edge_length(g[3] * kDeg2Rad, edge);
        const float wp = edge_length(p[2] * kDeg2Rad, edge), hp = edge_length(p[3] * kDeg2Rad, edge);
        dg = fmaf(wg, wg, hg * hg);
        dp = fmaf(wp, wp, hp * hp);
        quarter = 0.25f;
    }
    const float prod = dg * dp;
    // + 9.01e-3: the 1.5e-3 rad margin (both jitters + the reference's rounding of A) moved into R^2 (header comment)
    const float R2 = fmaf(quarter, fmaf(2.0f, prod * fast_rsq(prod), dg + dp), 9.01e-3f);
    // 2 cos(angular distance) = cos(phi_g - phi_p) (1 + cos dtheta) + cos(phi_g + phi_p) (1 - cos dtheta)
    const float kRev = 1.0f / 360.0f;
    const float ag = g[1] * kRev;
    const float u = hw_cos_rev(fmaf(p[1], -kRev, ag)), v = hw_cos_rev(fmaf(p[1], kRev, ag));
    const float cD = hw_cos_rev((p[0] - g[0]) * kRev);
    float C2 = fmaf(u - v, cD, u + v);
    if (DIM == 5) C2 = fmaf(g[4] + p[4], 0.0f, C2);   // a NaN / infinite gamma must not be culled either
    // against twice the lower bound of cos R (1 - R^2/2 + R^4/24 - R^6/720 <= cos R), less 1e-4 for the hardware trig.
    // The bound decreases in R^2 on the whole axis (its derivative -1/2 + x/12 - x^2/240 has no real root) and is below
    // -1.12 from R^2 = 8.9 on, where 2 cos >= -2 (1 + 3e-6) can never be below twice it: no separate test of R.
    // CHORD (sph2pob_legacy): cos L < 1 - R^2/2, which is below -1 from R^2 = 4 on
    const float bound2 = CHORD ? (2.0f - 2e-4f) - R2
                               : fmaf(fmaf(fmaf(-1.0f / 360.0f, R2, 1.0f / 12.0f), R2, -1.0f), R2, 2.0f - 2e-4f);
    return in_range & (C2 < bound2);
}

// trig by-products of stage 1 that the loss adjoint reuses
struct FastTrig { float sg, cg, sp, cp, sD, cD; };

// The two planar boxes after the rotated jitter, in the frame "P at the origin": T's centre is (dx, dy); angles as
// (cos, sin); (c, s) = cos / sin of (a_g - a_p).  g_* are gradient gates (false where a clamp / acos floor of the
// reference is active) — only filled in when lean_front is asked for them.
struct PlanarPair {
    float dx, dy, ca, sa, cb, sb, wg, hg, wp, hp, c, s;
    bool g_A, g_ag, g_ap, g_wg, g_hg, g_wp, g_hp;
};

// RBFoV `efficient`: the difference of the two planar angles a = floor(atan2(N, D)) - gamma as the reference forms it (not
// wrapped), for the rotated jitter's decisions on pairs whose directions are within 1.75e-3 of each other.  A called
// function (two atan2 that a few passes in a hundred need).
__host__ __device__ __attribute__((noinline)) inline float unwrapped_angle_diff(float D, float N, float Dp, float Np, float iS, float S2,
                                                                             float ga, float gb) {
    float c1 = D * iS, s1 = N * iS, c2 = Dp * iS, s2 = Np * iS;
    if (!(S2 > 1e-30f)) { c1 = 0.0f; s1 = 1.0f; c2 = 0.0f; s2 = 1.0f; }
    angle_floor(c1, s1);
    angle_floor(c2, s2);
    return (atan2_r(s1, c1) - ga) - (atan2_r(s2, c2) - gb);
}

// |gamma| beyond 177 deg (outside any coder's range): the rotated jitter's angle clamp to [-2pi + 2ea, 2pi - ea] /
// [-2pi + ea, 2pi - 2ea] (sph_iou_api.py:239-240) may act on a = atan2(.) - gamma.  Returns the clamped directions and
// whether either angle moved (then the gradient gate of that angle closes).
struct WideGamma { float ca, sa, cb, sb; bool changed, keep_g, keep_p; };
template <int VARIANT>
__host__ __device__ __attribute__((noinline)) inline WideGamma wide_gamma_clamp(float ca, float sa, float cb, float sb, float ga, float gb) {
    const float ea = (float)kEpsA;
    const float twopi = 6.283185307179586f;
    float a1 = atan2_r(sa, ca), a2 = atan2_r(sb, cb);  // wrapped representatives
    if (VARIANT == VARIANT_EFFICIENT) {                // un-wrap to the reference's real value (|a| < 3*pi)
        a1 += twopi * rintf((-ga - a1) / twopi);
        a2 += twopi * rintf((-gb - a2) / twopi);
    }
    const float k1 = fminf(fmaxf(a1, -twopi + 2.0f * ea), twopi - ea), k2 = fminf(fmaxf(a2, -twopi + ea), twopi - 2.0f * ea);
    WideGamma r{ca, sa, cb, sb, k1 != a1 || k2 != a2, k1 == a1, k2 == a2};
    if (r.changed) {
        sincos_r(k1 - twopi * rintf(k1 / twopi), r.sa, r.ca);
        sincos_r(k2 - twopi * rintf(k2 / twopi), r.sb, r.cb);
    }
    return r;
}

// "any lane of the wave": the rare branches of the finishing stage are guarded wave-uniformly, so a wave none of whose
// lanes needs a branch skips its code with one scalar branch instead of executing it under an empty mask or paying
// for always-on selects.  On the host (unit tests) a "wave" is one pair.
// What the guards cost (tools/ubench/finish_rate.hip, 8 resident waves per SIMD, survivors of the benchmark distribution:
// profiles/r02n_finish_rate*.log): 531 ns per pass per SIMD with them, 402 ns for the same arithmetic with every guarded
// block compiled out — each guard ends a basic block at a branch that waits for a VALU compare.  The alternatives cost
// more: always-on selects (round 1), and a straight-line common path that flags rare lanes and ends in ONE guard in front
// of a re-evaluation by this general form: 0.72 % of the survivors are rare (0.63 % alone have planar angles within
// 2e-3 of each other: the bearings of a pair are correlated), i.e. 31 % of the 51-lane passes would run twice — built
// and measured: 10.4 us against 8.4 us per 1 M pairs (profiles/r02o_ab_two_tier_*.log).
#if defined(SPH_ANY_LANE)
// (a microbenchmark's own definition: tools/ubench/finish_rate.hip -DNO_GUARDS prices the guards)
#elif defined(__HIP_DEVICE_COMPILE__)
// (Hand it ONE compare where possible and combine guards with ||: the mask of a single v_cmp is the ballot, whereas a
// condition built from several compares is first turned into 0 / 1 per lane and compared again, two more VALU
// instructions.  __builtin_expect: the rare blocks are laid out behind the loop, the common path falls through.)
#define SPH_ANY_LANE(cond) __builtin_expect(__builtin_amdgcn_ballot_w64(cond) != 0ull, 0)
#else
#define SPH_ANY_LANE(cond) (cond)
#endif
// diagnostic: -DSPH_GUARDS_OFF=<bit mask> compiles the guarded block of site k out (tools/ubench/finish_rate.hip prices
// the sites one by one); 0 in every build of the library
#if !defined(SPH_GUARDS_OFF)
#define SPH_GUARDS_OFF 0
#endif
#define SPH_SITE(k) (((SPH_GUARDS_OFF) >> (k)) & 1) ? false :
#if !defined(SPH_SITE_HIT)
#define SPH_SITE_HIT(k)   // diagnostic: counts how often a wave enters guarded block k (tools/ubench/finish_rate.hip -DCOUNT_SITES)
#endif

// Stage 1 + planar boxes for one pair of spherically jittered boxes (degrees): the closed-form front end shared by the
// IoU kernels (lean_finish) and the loss kernels (loss_front_fast) — one source for the planar pair, so that the IoU the
// loss differentiates is the IoU the assigner sees.
//
// The COMMON path is straight-line code: everything the reference only does for a few pairs in a thousand — the acos
// floors, the rotated jitter's decisions on real angles (2 atan2) and its bumps, the out-of-range gamma clamp — sits
// behind a wave-uniform guard.  Measured against the alternatives on MI355X (profiles/r02b_ablation_*.log, DESIGN.md
// §9): always-on selects (round 1) 9.0 us per 1 M pairs; a lean path with rare lanes re-run by a general form in place
// 12.1 us (0.7 % of the lanes, but every third wave holds one); the same deferred to an index stack and finished per
// workgroup 11.0 us (one latency-bound pass per workgroup at the tail); a separating-axis reject + second record stack
// in front of the clip 8.8 us with rare lanes dropped.
// Arithmetic: explicit FMAs, contraction off (see rot) — the same bits from every inlined copy.
// CLAMPED: the boxes went through jitter_spherical (phi in [0, 180], theta in [0, 360] degrees), which lets the three
// sincos calls use the cheap quadrant logic (bit-identical on that range).  GATES: also fill in the gradient gates.
// sin / cos of a box's jittered colatitude, computed once per box where a kernel pairs one box with many (the assigner's
// tiles): role 1 = a row of bboxes1, role 2 = a row of bboxes2 (the spherical jitter clamps the two roles differently)
struct ColatTrig { float s, c; };
SPH_DEV ColatTrig colat_trig(float phi_deg, int role) {
    const float lo = role == 1 ? (float)(2 * kEpsS) : (float)kEpsS, hi = role == 1 ? (float)(180.0 - kEpsS) : (float)(180.0 - 2 * kEpsS);
    ColatTrig t;
    sincos_colat(clampf(phi_deg, lo, hi) * kDeg2Rad, t.s, t.c);
    return t;
}

// PRE = 1: the colatitude trig of the two boxes is handed in (`pre1`, `pre2`: exactly what the function would compute);
// PRE = 2: that of the first box only (one box against many: an NMS row)
template <int VARIANT, int DIM, bool GATES, bool CLAMPED = true, int PRE = 0>
SPH_DEV void lean_front(const float (&x1)[5], const float (&x2)[5], int edge, PlanarPair& o, FastTrig* trig = nullptr,
                        ColatTrig pre1 = ColatTrig{0.0f, 1.0f}, ColatTrig pre2 = ColatTrig{0.0f, 1.0f}) {
    const float e = (float)kEpsS, ea = (float)kEpsA;
    // ---- stage 1: accurate trig on the jittered boxes, bearing numerators; degrees -> radians with the reference's
    // rounding (torch.deg2rad: x * fl32(pi/180)) ----
    const float thg = x1[0] * kDeg2Rad, phg = x1[1] * kDeg2Rad, thp = x2[0] * kDeg2Rad, php = x2[1] * kDeg2Rad;
    float wg = edge_length(x1[2] * kDeg2Rad, edge), hg = edge_length(x1[3] * kDeg2Rad, edge);
    float wp = edge_length(x2[2] * kDeg2Rad, edge), hp = edge_length(x2[3] * kDeg2Rad, edge);
    float sg, cg, sp, cp, sD, h2;   // sD = sin(theta_p - theta_g), h2 = 1 - cos(theta_p - theta_g)
    if (CLAMPED) {
        if (PRE) { sg = pre1.s; cg = pre1.c; } else sincos_colat(phg, sg, cg);
        if (PRE == 1) { sp = pre2.s; cp = pre2.c; } else sincos_colat(php, sp, cp);
        sin_vers_double(0.5f * (thp - thg), sD, h2);
    } else {
        float sh, ch;
        sincos_r(phg, sg, cg);
        sincos_r(php, sp, cp);
        sincos_r(0.5f * (thp - thg), sh, ch);
        const float t = sh * ch;
        sD = t + t;
        h2 = 2.0f * sh * sh;
    }
    const float spcg = sp * cg, sgcp = sg * cp, sgsp = sg * sp;
    const float q = spcg - sgcp;                     // sin(phi_p - phi_g)
    const float N = fmaf(-spcg, h2, q), D = -sp * sD;
    const float Np = fmaf(sgcp, h2, q), Dp = -sg * sD;
    const float C = fmaf(-sgsp, h2, fmaf(cg, cp, sgsp));
    if (trig) { trig->sg = sg; trig->cg = cg; trig->sp = sp; trig->cp = cp; trig->sD = sD; trig->cD = 1.0f - h2; }
    // ---- planar boxes as (cos, sin) ----
    const float S2 = fmaf(N, N, D * D);
    const float iS = fast_rsq(S2);
    float A = atan2_pos(S2 * iS, C);
    const float Amin = VARIANT == VARIANT_STANDARD ? 2.0f * kMinAng : kMinAng;
    if (GATES) o.g_A = A > Amin;
    A = fmaxf(A, Amin);
    float ca = D * iS, sa = N * iS, cb = Dp * iS, sb = Np * iS;
    // exactly coincident (or exactly antipodal) centres: the bearing is undefined (0/0).  N and D are products, not
    // differences, so they stay meaningful down to ~1e-15 (e.g. two boxes clamped onto a pole: A ~ 1e-7 but the bearings
    // still differ by the longitude difference); only a literal zero needs a convention: a = pi/2.
    const bool coincident = !(S2 > 1e-30f);
    auto fix_coincident = [&]() {
        if (coincident) { ca = 0.0f; sa = 1.0f; cb = 0.0f; sb = 1.0f; }
    };
    float ga = 0.0f, gb = 0.0f;
    // the reference's sign * |acos(clamp(cos a))| floors |a| and |pi - a| at kMinAng (angle_floor).  ONE guard for the
    // coincident-centre fix and the floors where the floors come first (every guard ends a basic block at a branch that
    // waits for its compare: see SPH_ANY_LANE); a coincident pair's sines are NaN or inf, so its lane needs no floor
    // compare of its own before the fix, and after it (sin = 1) none applies.
    auto floors = [&](bool with_coincident) {
        const bool fa = fabsf(sa) < kMinAng, fb = fabsf(sb) < kMinAng;
        if (SPH_SITE(1) ((with_coincident && SPH_ANY_LANE(coincident)) || SPH_ANY_LANE(fa) || SPH_ANY_LANE(fb))) {
            SPH_SITE_HIT(1);
            bool fa2 = fa, fb2 = fb;
            if (with_coincident) { fix_coincident(); fa2 = fabsf(sa) < kMinAng; fb2 = fabsf(sb) < kMinAng; }
            if (GATES) { o.g_ag = !fa2; o.g_ap = !fb2; }
            angle_floor(ca, sa);
            angle_floor(cb, sb);
        } else if (GATES) { o.g_ag = true; o.g_ap = true; }
    };
    if (DIM == 5) {
        ga = x1[4] * kDeg2Rad;
        gb = x2[4] * kDeg2Rad;
        float sga, cga, sgb, cgb;
        sincos_r(ga, sga, cga);
        sincos_r(gb, sgb, cgb);
        if (VARIANT == VARIANT_EFFICIENT) floors(true);   // floor, then a -= gamma
        else if (SPH_ANY_LANE(coincident)) fix_coincident();
        rot(ca, sa, cga, -sga);
        rot(cb, sb, cgb, -sgb);
        if (VARIANT == VARIANT_STANDARD) floors(false);    // d rotated first
    } else {
        floors(true);
    }
    // ---- rotated jitter (sph_iou_api.py:222-242) on (x, w, h, a); its decisions need real angles only when the two
    // angles are within ~1.8e-3 of each other modulo 2 pi ----
    float c = fmaf(ca, cb, sa * sb), s = fmaf(sa, cb, -(ca * sb));  // cos / sin of (a_g - a_p)
    const bool sim_dist = A < e, sim_size = fminf(fabsf(wg - wp), fabsf(hg - hp)) < e;
    bool sim = sim_dist | sim_size;
    bool close = false;
    // candidates for the angle decisions: `close` needs |a1 - a2| < ea = 1.2346e-3 after a possible shift of the difference
    // by 4e = 4.94e-4 (`sim`), i.e. |a_g - a_p| < 1.729e-3 — and 0.54 % of the benchmark's survivors are within 2e-3 (the
    // two bearings of a pair are correlated), which sends every third 64-lane pass through this block
    const bool cand_s = fabsf(s) < 1.75e-3f, cand = (c > 0.5f) & cand_s;
    float dx = A, dy = 0.0f;
    // ONE guard for the decisions and the adjustments (the wave-level test of the candidates is on the sine alone: a
    // single compare's mask, see SPH_ANY_LANE)
    if (SPH_SITE(2) (SPH_ANY_LANE(cand_s) || SPH_ANY_LANE(sim_dist) || SPH_ANY_LANE(sim_size))) {
        SPH_SITE_HIT(2);
        if (cand) {
            float dang;   // a_g - a_p as the reference forms it: the difference of two angles in [-pi, pi], not wrapped
            if (DIM == 5 && VARIANT == VARIANT_EFFICIENT) {  // a = floor(atan2(N, D)) - gamma, not wrapped
                dang = unwrapped_angle_diff(D, N, Dp, Np, iS, S2, ga, gb);
            } else {
#if defined(SPH_ATAN_DECISIONS)
                dang = atan2_r(sa, ca) - atan2_r(sb, cb);
#else
                // from the sine of the difference (|s| < 1.75e-3, c > 0.5: asin s = s + s^3 / 6 to 1e-17), which resolves
                // the thresholds to ~1e-10 where the difference of two rounded angles near pi carries ~5e-7; the two
                // angles are on opposite sides of the +-pi cut — their difference is ~2 pi, no decision fires — exactly
                // when both cosines are negative and the sines differ in sign (the floors keep |sin| >= 4.88e-4)
                const bool straddle = (ca < 0.0f) & ((sa < 0.0f) != (sb < 0.0f));
                dang = straddle ? 6.0f : fmaf(s * s, s * (1.0f / 6.0f), s);
#endif
            }
            sim |= fabsf(dang) < e;
            if (sim) dang -= (float)(4 * kEpsS);   // a_g += e, a_p += 5 e
            close = fabsf(dang) < ea;
        }
        // constant rotations of (cos, sin) instead of new trig
        if (sim) {
            dx += e; dy += e;  // (x, y) += (e, e) vs (2e, 2e)
            wg += (float)(2 * kEpsS); hg += (float)(2 * kEpsS); wp += e; hp += e;
            rot(ca, sa, (float)0.99999999237921, (float)1.2345678e-4);    // cos/sin(e)
            rot(cb, sb, (float)0.99999980948025, (float)6.172838610e-4);  // cos/sin(5e)
        }
        if (close) {
            rot(ca, sa, (float)0.99999923792122, (float)1.2345674864e-3);  // cos/sin(ea)
            rot(cb, sb, (float)0.99999695168547, (float)2.4691330913e-3);  // cos/sin(2ea)
        }
        if (sim | close) {   // only the lanes whose angles moved: the others keep their bits whatever the wave holds
            c = fmaf(ca, cb, sa * sb);
            s = fmaf(sa, cb, -(ca * sb));
        }
    }
    if (GATES) {
        o.g_wg = wg >= (float)(2 * kEpsA / 10); o.g_hg = hg >= (float)(2 * kEpsA / 10);
        o.g_wp = wp >= (float)(kEpsA / 10);     o.g_hp = hp >= (float)(kEpsA / 10);
    }
    // (one v_med3 each: fmaxf on a value that reaches here through a guarded block costs a v_max x, x in front of it)
    wg = clampf(wg, (float)(2 * kEpsA / 10), 3.0e38f); hg = clampf(hg, (float)(2 * kEpsA / 10), 3.0e38f);
    wp = clampf(wp, (float)(kEpsA / 10), 3.0e38f);     hp = clampf(hp, (float)(kEpsA / 10), 3.0e38f);
    if (DIM == 5) {
        const bool wide = (fabsf(ga) > 3.1f) | (fabsf(gb) > 3.1f);
        if (SPH_ANY_LANE(wide)) {
            if (wide) {   // a called function, like near_parallel_inter: ~200 instructions no coder's boxes ever reach
                const WideGamma r = wide_gamma_clamp<VARIANT>(ca, sa, cb, sb, ga, gb);
                if (r.changed) {
                    ca = r.ca; sa = r.sa; cb = r.cb; sb = r.sb;
                    c = fmaf(ca, cb, sa * sb);
                    s = fmaf(sa, cb, -(ca * sb));
                    if (GATES) { o.g_ag &= r.keep_g; o.g_ap &= r.keep_p; }
                }
            }
        }
    }
    o.dx = dx; o.dy = dy; o.ca = ca; o.sa = sa; o.cb = cb; o.sb = sb; o.c = c; o.s = s;
    o.wg = wg; o.hg = hg; o.wp = wp; o.hp = hp;
}

// Spherical jitter + stages 1 + 2 for one pair that survived the cull: clamp(IoU, 0, 1).  The spherical jitter's shift
// and the near-parallel safeguard are guarded like lean_front's rare branches.  A NaN coordinate gives NaN, as the
// reference's torch.clamp chain does (sph_iou_api.py:86, :244-260).
template <int VARIANT, int DIM, int PRE = 0>
SPH_DEV float lean_finish(const float (&in1)[5], const float (&in2)[5], int mode, int edge, ColatTrig pre1 = ColatTrig{0.0f, 1.0f},
                          ColatTrig pre2 = ColatTrig{0.0f, 1.0f}) {
    const float e = (float)kEpsS, e2 = (float)(2 * kEpsS);
    // evaluated HERE, not where it is used: left to the scheduler the test sinks to the end of the pass and keeps the
    // eight raw coordinates alive (and copied) through all of it
    float carrier = pair_nan_carrier<DIM>(in1, in2);
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(carrier));
#endif
    // ---- jitter_spherical (sph_iou_api.py:244-260): shift only where `similar`, clamps always ----
    float x1[5], x2[5];
    bool similar = false;
#pragma unroll
    for (int k = 0; k < 5; k++) { x1[k] = in1[k]; x2[k] = in2[k]; }
#pragma unroll
    for (int k = 0; k < DIM; k++) similar |= fabsf(in1[k] - in2[k]) < e;
    if (SPH_SITE(0) SPH_ANY_LANE(similar)) {
        SPH_SITE_HIT(0);
        const float sh1 = similar ? e2 : 0.0f, sh2 = similar ? e : 0.0f;  // x - 0 == x exactly
#pragma unroll
        for (int k = 0; k < DIM; k++) { x1[k] = x1[k] - sh1; x2[k] = x2[k] + sh2; }
        if (PRE && similar) { pre1 = colat_trig(x1[1], 1); if (PRE == 1) pre2 = colat_trig(x2[1], 2); }   // the shifted colatitudes
    }
    x1[0] = clampf(x1[0], e2, (float)(360.0 - kEpsS));
    x2[0] = clampf(x2[0], e, (float)(360.0 - 2 * kEpsS));
#pragma unroll
    for (int k = 1; k < 4; k++) {
        x1[k] = clampf(x1[k], e2, (float)(180.0 - kEpsS));
        x2[k] = clampf(x2[k], e, (float)(180.0 - 2 * kEpsS));
    }
    if (DIM == 5) x2[4] = clampf(x2[4], (float)(-360.0 + 2 * kEpsS), (float)(360.0 - 2 * kEpsS));  // the two clamps of :256-258
    PlanarPair q;
    lean_front<VARIANT, DIM, false, true, PRE>(x1, x2, edge, q, nullptr, pre1, pre2);
    // ---- stage 2: boundary integral of the two rectangles (P at the origin, T at (dx, dy)) ----
    const float kBig = 1e18f;
    const float ic = fminf(fmaxf(fast_rcp(q.c), -kBig), kBig), is = fminf(fmaxf(fast_rcp(q.s), -kBig), kBig);
    const float aic = fabsf(ic), ais = fabsf(is);
    const float hwa = 0.5f * q.wg, hha = 0.5f * q.hg, hwb = 0.5f * q.wp, hhb = 0.5f * q.hp;
    const float pax = -fmaf(q.dx, q.cb, q.dy * q.sb), pay = fmaf(q.dx, q.sb, -(q.dy * q.cb));
    const float pbx = fmaf(q.dx, q.ca, q.dy * q.sa), pby = fmaf(q.dy, q.ca, -(q.dx * q.sa));
    float t2 = edges_inside3(pax, pay, q.c, q.s, ic, is, aic, ais, hwa, hha, hwb, hhb, q.wg, q.hg, true) +
               edges_inside3(pbx, pby, q.c, -q.s, ic, -is, aic, ais, hwb, hhb, hwa, hha, q.wp, q.hp, false);
    // the two jitter steps cancelled: DESIGN.md §9.  (Two compares and a scalar OR of their masks; written as one
    // condition the compiler turns them into abs / canonicalise / min / compare: five VALU instructions)
    const bool near_s = fabsf(q.s) < kNearParallel, near_c = fabsf(q.c) < kNearParallel;
    if (SPH_SITE(3) (SPH_ANY_LANE(near_s) || SPH_ANY_LANE(near_c))) {
        SPH_SITE_HIT(3);
        if (near_s | near_c) t2 = 2.0f * near_parallel_inter(pax, pay, q.c, q.s, hwa, hha, hwb, hhb);
    }
    const float inter = 0.5f * clampf(t2, 0.0f, 3.0e38f);
    const float a1 = q.wg * q.hg, a2 = q.wp * q.hp;
    const float base = mode == MODE_IOU ? (a1 + a2 - inter) : a1;
    float rb = fast_rcp(base);
    rb = rb * fmaf(-base, rb, 2.0f);  // one Newton step: ~0.5 ulp quotient without the IEEE divide expansion
    const float iou = fminf(fmaxf(inter * rb, 0.0f), 1.0f);
    return carrier != carrier ? __builtin_nanf("") : iou;
}

// VARIANT: 0 standard, 1 efficient.  Returns clamp(IoU, 0, 1) of one pair (one-lane-per-pair kernels, NMS rows, host
// tests): the compacting kernels run exactly these two functions, hence bit-identical results everywhere.
template <int VARIANT, int DIM>
SPH_DEV float pair_iou_fast(const float (&in1)[5], const float (&in2)[5], int mode, int edge) {
    if (fast_cull<DIM>(in1, in2, edge)) return 0.0f;
    return lean_finish<VARIANT, DIM>(in1, in2, mode, edge);
}

}  // namespace sph2pob

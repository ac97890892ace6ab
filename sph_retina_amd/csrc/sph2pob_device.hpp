// Sph2Pob device math for gfx950 (CDNA4).  One lane = one box pair; everything lives in VGPRs.
//
// What it computes (reference paths are relative to the reference checkout):
//   jitter_spherical   sphdet/iou/sph_iou_api.py:244-260
//   transform_*        sphdet/iou/sph2pob_standard.py:8-80, sph2pob_efficient.py:9-73, sph2pob_legacy.py:8-31
//   jitter_rotated     sphdet/iou/sph_iou_api.py:222-242
//   rect_intersection  replaces mmcv-full 1.6.0 box_iou_rotated (call site sph_iou_api.py:79)
//
// The transforms keep the reference's fp32 operation order (this translation unit is compiled with
// -ffp-contract=off): the path is dominated by acos(clamp(dot)) of nearly parallel unit vectors, which
// amplifies any re-association into >1e-5 IoU differences at small centre distances.
//
// The planar stage is NOT mmcv's algorithm (24 candidate points -> Graham scan with absolute 1e-6/1e-8
// tolerances -> fan area: divergent, array-indexed, scratch-heavy on a GPU).  It is a branch-free boundary
// integral: the boundary of A∩B is (edges of A inside B) ∪ (edges of B inside A); each of the 8 edges is
// clipped against the other rectangle with a slab test in that rectangle's frame and contributes
// (clipped length) x (signed distance of the edge line from the integration origin).  No sort, no hull, no
// local arrays, IEEE inf/NaN semantics of v_rcp/v_min/v_max handle parallel edges.
#pragma once
#include <hip/hip_runtime.h>

namespace sph2pob {

enum : int { VARIANT_STANDARD = 0, VARIANT_EFFICIENT = 1, VARIANT_LEGACY = 2, VARIANT_SPH_IOU = 3, VARIANT_FOV_IOU = 4,
             VARIANT_UNBIASED = 5, VARIANT_NAIVE = 6 };
enum : int { MODE_IOU = 0, MODE_IOF = 1 };
enum : int { EDGE_ARC = 0, EDGE_CHORD = 1, EDGE_TANGENT = 2 };
enum : int { ANGLE_EQUATOR = 0, ANGLE_PROJECT = 1 };

// The reference-order transforms below are written once for a scalar type T: float — the kernels' arithmetic, the
// reference's own fp32 operation order — and Dual, a forward-mode (value, derivative) pair used ONLY by the adjoint of
// the transforms that have no closed-form backward here (sph2pob_legacy, rbb_angle='project': transform_bwd_dual_kernel).
// With T = float every helper is the plain libm / builtin call it replaces: same instructions, same bits.
struct Dual { float v, d; };
template <class T> struct PBoxT { T x, y, w, h, a; };
template <class T> struct V3T { T x, y, z; };
using PBox = PBoxT<float>;
using V3 = V3T<float>;

#define SPH_DEV __host__ __device__ __forceinline__

// ---- constants, rounded to fp32 exactly like the reference's python-double -> float32 casts ----
constexpr float kDeg2Rad = 0.017453292519943295f;       // torch.deg2rad scalar
constexpr float kPi = 3.141592653589793f;               // torch.pi cast to fp32
constexpr double kEpsS = 1e-4 * 1.2345678;              // spherical / rotated "similar" eps
constexpr double kEpsA = 1e-3 * 1.2345678;              // rotated angle eps
constexpr float kClampHi = (float)(1 - 1e-7);           // 0.99999988
constexpr float kClampLo = (float)(-1 + 1e-7);

// clamp for NaN-free x: a single v_med3_f32 on the device
SPH_DEV float clampf(float x, float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(x, lo, hi);
#else
    return fminf(fmaxf(x, lo), hi);
#endif
}

// NaN-propagating min / max (IEEE-754-2019 minimum / maximum: one v_minimum3_f32 / v_maximum3_f32 on gfx950, and no
// v_max x, x canonicalisation in front of them).  torch.clamp propagates NaN and maps +-inf to the bound; fminf / fmaxf
// and v_med3_f32 return the non-NaN operand, which would turn a NaN box of a diverged network into a valid-looking one.
SPH_DEV float min_nan(float a, float b) { return __builtin_elementwise_minimum(a, b); }
SPH_DEV float max_nan(float a, float b) { return __builtin_elementwise_maximum(a, b); }
// NaN iff any coordinate of the pair is NaN (+-inf stays inf: the reference's clamps make it finite; an infinite gamma,
// which the reference never clamps for bboxes1 and whose sine is NaN, counts as NaN)
// the NaN-propagating maximum of the pair's coordinates: NaN iff any of them is (a chain of three-input maxima: four
// v_maximum3_f32 for the eight coordinates of a BFoV pair)
template <int DIM>
SPH_DEV float pair_nan_carrier(const float (&a)[5], const float (&b)[5]) {
    float m = max_nan(max_nan(a[0], a[1]), a[2]);
    m = max_nan(max_nan(m, a[3]), b[0]);
    m = max_nan(max_nan(m, b[1]), b[2]);
    m = max_nan(m, b[3]);
    if (DIM == 5) m = max_nan(max_nan(m, a[4] - a[4]), b[4] - b[4]);   // inf - inf = NaN
    return m;
}
template <int DIM>
SPH_DEV bool pair_has_nan(const float (&a)[5], const float (&b)[5]) {
    const float m = pair_nan_carrier<DIM>(a, b);
    return m != m;
}

// ---- Dual arithmetic (torch autograd's rules: clamp / max pass the gradient on the closed side, |x| uses sign) ----
SPH_DEV float val(float x) { return x; }
SPH_DEV float val(Dual x) { return x.v; }
SPH_DEV Dual operator+(Dual a, Dual b) { return Dual{a.v + b.v, a.d + b.d}; }
SPH_DEV Dual operator-(Dual a, Dual b) { return Dual{a.v - b.v, a.d - b.d}; }
SPH_DEV Dual operator-(Dual a) { return Dual{-a.v, -a.d}; }
SPH_DEV Dual operator*(Dual a, Dual b) { return Dual{a.v * b.v, a.d * b.v + a.v * b.d}; }
SPH_DEV Dual operator/(Dual a, Dual b) { const float q = a.v / b.v; return Dual{q, (a.d - q * b.d) / b.v}; }
SPH_DEV Dual operator+(Dual a, float b) { return Dual{a.v + b, a.d}; }
SPH_DEV Dual operator+(float a, Dual b) { return Dual{a + b.v, b.d}; }
SPH_DEV Dual operator-(Dual a, float b) { return Dual{a.v - b, a.d}; }
SPH_DEV Dual operator-(float a, Dual b) { return Dual{a - b.v, -b.d}; }
SPH_DEV Dual operator*(Dual a, float b) { return Dual{a.v * b, a.d * b}; }
SPH_DEV Dual operator*(float a, Dual b) { return Dual{a * b.v, a * b.d}; }
SPH_DEV Dual operator/(Dual a, float b) { return Dual{a.v / b, a.d / b}; }
SPH_DEV float m_sin(float x) { return sinf(x); }
SPH_DEV float m_cos(float x) { return cosf(x); }
SPH_DEV float m_tan(float x) { return tanf(x); }
SPH_DEV float m_acos(float x) { return acosf(x); }
SPH_DEV float m_asin(float x) { return asinf(x); }
SPH_DEV float m_sqrt(float x) { return sqrtf(x); }
SPH_DEV float m_abs(float x) { return fabsf(x); }
SPH_DEV float m_max(float x, float lo) { return fmaxf(x, lo); }
SPH_DEV float m_clamp(float x, float lo, float hi) { return clampf(x, lo, hi); }
SPH_DEV float m_fmod(float x, float m) { return fmodf(x, m); }
SPH_DEV Dual m_sin(Dual x) { return Dual{sinf(x.v), cosf(x.v) * x.d}; }
SPH_DEV Dual m_cos(Dual x) { return Dual{cosf(x.v), -sinf(x.v) * x.d}; }
SPH_DEV Dual m_tan(Dual x) { const float t = tanf(x.v); return Dual{t, (1.0f + t * t) * x.d}; }
SPH_DEV Dual m_acos(Dual x) { return Dual{acosf(x.v), -x.d / sqrtf(1.0f - x.v * x.v)}; }
SPH_DEV Dual m_asin(Dual x) { return Dual{asinf(x.v), x.d / sqrtf(1.0f - x.v * x.v)}; }
SPH_DEV Dual m_sqrt(Dual x) { const float r = sqrtf(x.v); return Dual{r, 0.5f * x.d / r}; }
SPH_DEV Dual m_abs(Dual x) { return Dual{fabsf(x.v), x.v > 0.0f ? x.d : (x.v < 0.0f ? -x.d : 0.0f)}; }
SPH_DEV Dual m_max(Dual x, float lo) { return Dual{fmaxf(x.v, lo), x.v >= lo ? x.d : 0.0f}; }
SPH_DEV Dual m_clamp(Dual x, float lo, float hi) { return Dual{clampf(x.v, lo, hi), (x.v >= lo && x.v <= hi) ? x.d : 0.0f}; }
SPH_DEV Dual m_fmod(Dual x, float m) { return Dual{fmodf(x.v, m), x.d}; }

template <class T> SPH_DEV V3T<T> v3(T x, T y, T z) { return V3T<T>{x, y, z}; }
template <class T> SPH_DEV V3T<T> operator+(V3T<T> a, V3T<T> b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class T> SPH_DEV V3T<T> operator-(V3T<T> a, V3T<T> b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class T> SPH_DEV T dot(V3T<T> a, V3T<T> b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
template <class T> SPH_DEV V3T<T> cross(V3T<T> a, V3T<T> b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// F.normalize: v / max(||v||, 1e-12)
template <class T> SPH_DEV V3T<T> normalize(V3T<T> a) {
    T n = m_sqrt((a.x * a.x + a.y * a.y) + a.z * a.z);
    T d = m_max(n, 1e-12f);
    return v3(a.x / d, a.y / d, a.z / d);
}
// compute_angle_between_direction (radians): sph2pob_efficient.py:192-208
template <class T> SPH_DEV T angle_between(V3T<T> a, V3T<T> b) {
    T c = m_clamp(dot(normalize(a), normalize(b)), kClampLo, kClampHi);
    return m_abs(m_acos(c));
}
template <class T> SPH_DEV T rad2deg_ref(T r) { return r / kPi * 180.0f; }  // sph2pob_standard.py:216
// compute_clockwise_or_anticlockwise_between_direction: sph2pob_efficient.py:211-226
template <class T> SPH_DEV float sign_mask(V3T<T> a, V3T<T> b, V3T<T> ref) { return val(dot(cross(a, b), ref)) < 0.0f ? 1.0f : -1.0f; }

template <class T> SPH_DEV T edge_length(T fov, int edge) {  // sph2pob_standard.py:110-118
    if (edge == EDGE_ARC) return fov;
    if (edge == EDGE_TANGENT) return 2.0f * m_tan(fov / 2.0f);
    return 2.0f * m_sin(fov / 2.0f);
}

// ------------------------------------------------------------------------------------------------
// jiter_spherical_bboxes — sph_iou_api.py:244-260.  b1/b2 are register copies (inputs are never mutated).
template <int DIM, class T>
SPH_DEV void jitter_spherical(T (&b1)[5], T (&b2)[5]) {
    const float eps = (float)kEpsS, eps2 = (float)(2 * kEpsS);
    bool similar = false;
#pragma unroll
    for (int k = 0; k < DIM; k++) similar |= fabsf(val(b1[k]) - val(b2[k])) < eps;
    const float sh1 = similar ? eps2 : 0.0f, sh2 = similar ? eps : 0.0f;  // x - 0 == x exactly
#pragma unroll
    for (int k = 0; k < DIM; k++) {
        b1[k] = b1[k] - sh1;
        b2[k] = b2[k] + sh2;
    }
    b1[0] = m_clamp(b1[0], eps2, (float)(360.0 - kEpsS));
    b2[0] = m_clamp(b2[0], eps, (float)(360.0 - 2 * kEpsS));
#pragma unroll
    for (int k = 1; k < 4; k++) {
        b1[k] = m_clamp(b1[k], eps2, (float)(180.0 - kEpsS));
        b2[k] = m_clamp(b2[k], eps, (float)(180.0 - 2 * kEpsS));
    }
    if (DIM == 5) {  // quirk kept: only bboxes2's gamma is clamped (twice) :256-258
        b2[4] = m_clamp(b2[4], (float)(-360.0 + kEpsS), (float)(360.0 - 2 * kEpsS));
        b2[4] = m_clamp(b2[4], (float)(-360.0 + 2 * kEpsS), (float)(360.0 - kEpsS));
    }
}

// jiter_rotated_bboxes — sph_iou_api.py:222-242
template <class T>
SPH_DEV void jitter_rotated(PBoxT<T>& p1, PBoxT<T>& p2) {
    const float e = (float)kEpsS, e2 = (float)(2 * kEpsS), e5 = (float)(5 * kEpsS);
    bool similar = (fabsf(val(p1.x) - val(p2.x)) < e) | (fabsf(val(p1.w) - val(p2.w)) < e) | (fabsf(val(p1.h) - val(p2.h)) < e) |
                   (fabsf(val(p1.a) - val(p2.a)) < e);
    if (similar) {
        p1.x = p1.x + e;  p1.y = p1.y + e;  p1.w = p1.w + e2; p1.h = p1.h + e2; p1.a = p1.a + e;
        p2.x = p2.x + e2; p2.y = p2.y + e2; p2.w = p2.w + e;  p2.h = p2.h + e;  p2.a = p2.a + e5;
    }
    const float ea = (float)kEpsA, ea2 = (float)(2 * kEpsA);
    if (fabsf(val(p1.a) - val(p2.a)) < ea) {
        p1.a = p1.a + ea;
        p2.a = p2.a + ea2;
    }
    const double pi = 3.141592653589793;
    p1.w = m_max(p1.w, (float)(2 * kEpsA / 10)); p1.h = m_max(p1.h, (float)(2 * kEpsA / 10));
    p2.w = m_max(p2.w, (float)(kEpsA / 10));     p2.h = m_max(p2.h, (float)(kEpsA / 10));
    p1.a = m_clamp(p1.a, (float)(-2 * pi + 2 * kEpsA), (float)(2 * pi - kEpsA));
    p2.a = m_clamp(p2.a, (float)(-2 * pi + kEpsA), (float)(2 * pi - 2 * kEpsA));
}

// ------------------------------------------------------------------------------------------------
template <class T> struct SBoxT { T th, ph, al, be, ga, st, ct, sp, cp; V3T<T> c, d; };
using SBox = SBoxT<float>;

template <int DIM, class T>
SPH_DEV SBoxT<T> load_sbox(const T (&b)[5]) {  // sph2pob_standard.py:23-41, 121-172
    SBoxT<T> s;
    s.th = b[0] * kDeg2Rad; s.ph = b[1] * kDeg2Rad; s.al = b[2] * kDeg2Rad; s.be = b[3] * kDeg2Rad;
    s.ga = DIM == 5 ? b[4] * kDeg2Rad : b[4] * 0.0f;
    s.st = m_sin(s.th); s.ct = m_cos(s.th); s.sp = m_sin(s.ph); s.cp = m_cos(s.ph);
    s.c = v3(s.sp * s.ct, s.sp * s.st, s.cp);
    s.d = v3(s.cp * s.ct, s.cp * s.st, -s.sp);
    return s;
}

template <class T> struct M3T { V3T<T> r0, r1, r2; };
using M3 = M3T<float>;
template <class T> SPH_DEV V3T<T> mul(const M3T<T>& m, V3T<T> v) { return v3(dot(m.r0, v), dot(m.r1, v), dot(m.r2, v)); }
// compute_rotate_matrix(theta, phi): rows look, down, right — sph2pob_standard.py:239-261
template <class T> SPH_DEV M3T<T> rotate_matrix(T st, T ct, T sp, T cp) {
    return M3T<T>{v3(sp * ct, sp * st, cp), v3(cp * ct, cp * st, -sp), v3(st, -ct, ct * 0.0f)};
}
// dir <- T^T (Rx(gamma) T) dir — compute_gamma_matrix, sph2pob_standard.py:300-314 (matrix products kept)
template <class T> SPH_DEV V3T<T> apply_gamma(const SBoxT<T>& s, T gamma, V3T<T> dir) {
    M3T<T> Tm = rotate_matrix(s.st, s.ct, s.sp, s.cp);
    T sg = m_sin(gamma), cg = m_cos(gamma);
    // RT = Rx * T  (rows)
    M3T<T> RT;
    RT.r0 = Tm.r0;
    RT.r1 = v3((0.0f * Tm.r0.x + cg * Tm.r1.x) + (-sg) * Tm.r2.x, (0.0f * Tm.r0.y + cg * Tm.r1.y) + (-sg) * Tm.r2.y,
               (0.0f * Tm.r0.z + cg * Tm.r1.z) + (-sg) * Tm.r2.z);
    RT.r2 = v3((0.0f * Tm.r0.x + sg * Tm.r1.x) + cg * Tm.r2.x, (0.0f * Tm.r0.y + sg * Tm.r1.y) + cg * Tm.r2.y,
               (0.0f * Tm.r0.z + sg * Tm.r1.z) + cg * Tm.r2.z);
    // G = T^T * RT : G[i][j] = sum_k T[k][i] * RT[k][j]
    V3T<T> c0 = v3(Tm.r0.x, Tm.r1.x, Tm.r2.x), c1 = v3(Tm.r0.y, Tm.r1.y, Tm.r2.y), c2 = v3(Tm.r0.z, Tm.r1.z, Tm.r2.z);
    V3T<T> q0 = v3(RT.r0.x, RT.r1.x, RT.r2.x), q1 = v3(RT.r0.y, RT.r1.y, RT.r2.y), q2 = v3(RT.r0.z, RT.r1.z, RT.r2.z);
    M3T<T> Gm{v3(dot(c0, q0), dot(c0, q1), dot(c0, q2)), v3(dot(c1, q0), dot(c1, q1), dot(c1, q2)),
              v3(dot(c2, q0), dot(c2, q1), dot(c2, q2))};
    return mul(Gm, dir);
}

template <int DIM, class T>
SPH_DEV void transform_standard(const T (&g_)[5], const T (&p_)[5], int edge, int angle, PBoxT<T>& og, PBoxT<T>& op) {
    SBoxT<T> g = load_sbox<DIM>(g_), p = load_sbox<DIM>(p_);
    M3T<T> R;
    V3T<T> df = g.c - p.c;
    float l1 = (fabsf(val(df.x)) + fabsf(val(df.y))) + fabsf(val(df.z));
    if (l1 > 1e-8f) {  // compute_rotate_matrix_better :264-283
        V3T<T> look = normalize(g.c + p.c);
        V3T<T> right = normalize(p.c - g.c);
        R = M3T<T>{look, right, cross(look, right)};
    } else {  // compute_rotate_matrix(theta_r, phi_r) :286-297
        T th_r = (g.th + p.th) / 2.0f, ph_r = (g.ph + p.ph) / 2.0f;
        R = rotate_matrix(m_sin(th_r), m_cos(th_r), m_sin(ph_r), m_cos(ph_r));
    }
    V3T<T> dg = g.d, dp = p.d;
    if (DIM == 5) {
        dg = apply_gamma(g, -g.ga, dg);
        dp = apply_gamma(p, -p.ga, dp);
    }
    V3T<T> cg = mul(R, g.c), cp = mul(R, p.c);
    dg = mul(R, dg);
    dp = mul(R, dp);
    const T zero = g.th * 0.0f, one = zero + 1.0f;   // constants of the scalar type (derivative 0)
    const V3T<T> ez = v3(zero, zero, one), ex = v3(one, zero, zero);
    auto internal_angle = [&](V3T<T> d) {  // compute_internal_angle :88-108 (+ deg2rad of standardize_rotated_box)
        if (angle == ANGLE_PROJECT) d.x = zero;
        T a = m_abs(rad2deg_ref(angle_between(d, ez)));
        a = a * (val(d.y) > 0.0f ? 1.0f : -1.0f);  // sign_mask(ez, d, ex) == (-d.y < 0)
        return a * kDeg2Rad;
    };
    auto sph_coord = [&](V3T<T> c, T& th, T& ph) {  // compute_spherical_coordinate :175-199
        ph = rad2deg_ref(angle_between(c, ez)) * kDeg2Rad;
        V3T<T> cxy = v3(c.x, c.y, zero);
        T t = rad2deg_ref(angle_between(cxy, ex));
        t = t * (val(c.y) > 0.0f ? 1.0f : -1.0f);  // sign_mask(ex, cxy, -ez) == (-c.y < 0)
        th = t * kDeg2Rad;
    };
    og.a = internal_angle(dg);
    op.a = internal_angle(dp);
    sph_coord(cg, og.x, og.y);
    sph_coord(cp, op.x, op.y);
    og.w = edge_length(g.al, edge); og.h = edge_length(g.be, edge);
    op.w = edge_length(p.al, edge); op.h = edge_length(p.be, edge);
}

template <int DIM, class T>
SPH_DEV void transform_efficient(const T (&g_)[5], const T (&p_)[5], int edge, int angle, PBoxT<T>& og, PBoxT<T>& op) {
    SBoxT<T> g = load_sbox<DIM>(g_), p = load_sbox<DIM>(p_);
    V3T<T> z = cross(g.c, p.c);
    V3T<T> s = g.c + p.c;
    V3T<T> ref = v3(s.x / 2.0f, s.y / 2.0f, s.z / 2.0f);
    T arc = angle_between(g.c, p.c);
    V3T<T> dg = g.d, dp = p.d;
    const T zero = g.th * 0.0f;
    if (angle == ANGLE_PROJECT) { dg.x = zero; dp.x = zero; }  // :92-93, unrotated-frame quirk kept
    T ag = angle_between(dg, z) * sign_mask(z, dg, ref);
    T ap = angle_between(dp, z) * sign_mask(z, dp, ref);
    if (DIM == 5) { ag = ag - g.ga; ap = ap - p.ga; }
    og = PBoxT<T>{zero, zero, edge_length(g.al, edge), edge_length(g.be, edge), ag};
    op = PBoxT<T>{arc, zero, edge_length(p.al, edge), edge_length(p.be, edge), ap};
}

template <class T>
SPH_DEV T legacy_angle_aux(T th_box, T ph_box, T th_ref, T ph_ref) {  // sph2pob_legacy.py:120-134
    T sb = m_sin(th_box), cb = m_cos(th_box), spb = m_sin(ph_box), cpb = m_cos(ph_box);
    T sr = m_sin(th_ref), cr = m_cos(th_ref), spr = m_sin(ph_ref), cpr = m_cos(ph_ref);
    V3T<T> db = v3(cpb * cb, cpb * sb, -spb), dr = v3(cpr * cr, cpr * sr, -spr);
    T a = m_abs(rad2deg_ref(angle_between(db, dr)));
    const float hp = (float)(3.141592653589793 / 2);
    bool sign = ((val(th_box) >= val(th_ref)) && (val(ph_box) < hp)) || ((val(th_box) <= val(th_ref)) && (val(ph_box) > hp));
    return sign ? a : a * -1.0f;
}
template <class T>
SPH_DEV void transform_legacy(const T (&g_)[5], const T (&p_)[5], int edge, PBoxT<T>& og, PBoxT<T>& op) {
    T g0 = g_[0], p0 = p_[0];
    if (fabsf(val(g0) - val(p0)) > 180.0f) {  // standardize_spherical_box :236-257
        g0 = m_fmod(g0 + 180.0f, 360.0f);
        p0 = m_fmod(p0 + 180.0f, 360.0f);
    }
    const float hpi = (float)(3.141592653589793 / 2);
    T thg = g0 * kDeg2Rad - kPi, phg = hpi - g_[1] * kDeg2Rad;  // 'convention' :217-234
    T thp = p0 * kDeg2Rad - kPi, php = hpi - p_[1] * kDeg2Rad;
    T phi_i = (phg + php) / 2.0f;
    T phg_ = phg - phi_i, php_ = php - phi_i;
    T dphi = m_abs(phg - php), dth = m_abs(thg - thp);
    T s1 = m_sin(dphi / 2.0f), s2 = m_sin(dth / 2.0f);
    T L = 2.0f * m_asin(m_sqrt(s1 * s1 + (m_cos(phg) * m_cos(php)) * (s2 * s2)));  // :63-66
    T sl = m_sin(L / 2.0f);
    T q = (sl * sl - s1 * s1) / (m_cos(phg_) * m_cos(php_));
    T dth_ = m_abs(2.0f * m_asin(m_sqrt(q)));  // :70-72 (NaN for q < 0, as in the reference)
    float sgn = val(thp) > val(thg) ? 1.0f : -1.0f;
    T mg = g0 * kDeg2Rad, mp = p0 * kDeg2Rad, pg = g_[1] * kDeg2Rad, pp = p_[1] * kDeg2Rad;  // 'math'
    T mid = (mg + mp) / 2.0f;
    og.x = g0 * 0.0f;  og.y = phg_;
    op.x = dth_ * sgn; op.y = php_;
    og.w = edge_length(g_[2] * kDeg2Rad, edge); og.h = edge_length(g_[3] * kDeg2Rad, edge);
    op.w = edge_length(p_[2] * kDeg2Rad, edge); op.h = edge_length(p_[3] * kDeg2Rad, edge);
    og.a = legacy_angle_aux(mg, pg, mid, pg) * kDeg2Rad;
    op.a = legacy_angle_aux(mp, pp, mid, pp) * kDeg2Rad;
}

template <int VARIANT, int DIM, class T>
SPH_DEV void transform(const T (&g)[5], const T (&p)[5], int edge, int angle, PBoxT<T>& og, PBoxT<T>& op) {
    if (VARIANT == VARIANT_STANDARD) transform_standard<DIM>(g, p, edge, angle, og, op);
    else if (VARIANT == VARIANT_EFFICIENT) transform_efficient<DIM>(g, p, edge, angle, og, op);
    else if (VARIANT == VARIANT_LEGACY) transform_legacy(g, p, edge, og, op);
}

// Adjoint of transform<VARIANT, DIM> by forward-mode differentiation (one pass per input coordinate: 2 * DIM passes of
// the reference-order transform on Dual numbers): gin = J^T gout.  For the transforms without a closed-form backward
// (sph2pob_legacy, rbb_angle='project'); the standard / efficient equator transforms use pair_transform_bwd.
// `jitter`: the caller's transform ran jitter_spherical before and jitter_rotated after (Sph2PobTransfrom with
// 'sph2pob_legacy', sph2pob_transform.py:28-30): both only add constants and clamp, so on Dual numbers they shift the
// evaluation point and gate derivatives (torch.clamp: 1 on the closed interval) like torch's in-place ops do.
template <int VARIANT, int DIM>
SPH_DEV void transform_bwd_dual(const float (&in1)[5], const float (&in2)[5], const float (&g1)[5], const float (&g2)[5],
                                int edge, int angle, bool jitter, float (&gin1)[5], float (&gin2)[5]) {
#pragma unroll 1
    for (int k = 0; k < 2 * DIM; k++) {
        Dual x[5], y[5];
#pragma unroll
        for (int c = 0; c < 5; c++) {
            x[c] = Dual{in1[c], (k < DIM && c == k) ? 1.0f : 0.0f};
            y[c] = Dual{in2[c], (k >= DIM && c == k - DIM) ? 1.0f : 0.0f};
        }
        PBoxT<Dual> p1, p2;
        if (jitter) jitter_spherical<DIM>(x, y);
        transform<VARIANT, DIM>(x, y, edge, angle, p1, p2);
        if (jitter) jitter_rotated(p1, p2);
        const float acc = ((g1[0] * p1.x.d + g1[1] * p1.y.d) + (g1[2] * p1.w.d + g1[3] * p1.h.d) + g1[4] * p1.a.d) +
                          ((g2[0] * p2.x.d + g2[1] * p2.y.d) + (g2[2] * p2.w.d + g2[3] * p2.h.d) + g2[4] * p2.a.d);
        if (k < DIM) gin1[k] = acc; else gin2[k - DIM] = acc;
    }
#pragma unroll
    for (int c = DIM; c < 5; c++) { gin1[c] = 0.0f; gin2[c] = 0.0f; }
}

// ------------------------------------------------------------------------------------------------
// Clipped length of the segment P(tau) = (px, py) + tau * (ux, uy), tau in [0, len], inside the
// axis-aligned box |x| <= hx, |y| <= hy.  iux/iuy are 1/ux, 1/uy (inf when parallel: the products below
// then evaluate to -inf/+inf "no constraint" or to an empty interval, NaN (0*inf) is dropped by fmin/fmax).
SPH_DEV float clip_len(float px, float py, float iux, float iuy, float len, float hx, float hy) {
    float ax = (-hx - px) * iux, bx = (hx - px) * iux;
    float ay = (-hy - py) * iuy, by = (hy - py) * iuy;
    float lo = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), 0.0f);
    float hi = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), len);
    return fmaxf(hi - lo, 0.0f);
}

// Twice the boundary integral over the 4 edges of rectangle A (half extents hwa, hha, axes rotated by
// (c, s) = (cos, sin) of (angle_A - angle_B) in B's frame, centre (pax, pay) in B's frame) clipped to
// B = [-hwb, hwb] x [-hhb, hhb]; integration origin = B's centre.
SPH_DEV float edges_inside(float pax, float pay, float c, float s, float ic, float is, float hwa, float hha,
                           float hwb, float hhb, float wa, float ha, bool with_origin_terms) {
    float ux = hwa * c, uy = hwa * s;    // +u half extent
    float vx = -hha * s, vy = hha * c;   // +v half extent
    // corners k0 = +u+v, k1 = -u+v, k2 = -u-v, k3 = +u-v (counter-clockwise)
    float k0x = pax + ux + vx, k0y = pay + uy + vy;
    float k1x = pax - ux + vx, k1y = pay - uy + vy;
    float k2x = pax - ux - vx, k2y = pay - uy - vy;
    float k3x = pax + ux - vx, k3y = pay + uy - vy;
    // edge directions: e0 = -u (k0->k1), e1 = -v (k1->k2), e2 = +u (k2->k3), e3 = +v (k3->k0)
    float l0 = clip_len(k0x, k0y, -ic, -is, wa, hwb, hhb);
    float l1 = clip_len(k1x, k1y, is, -ic, ha, hwb, hhb);
    float l2 = clip_len(k2x, k2y, ic, is, wa, hwb, hhb);
    float l3 = clip_len(k3x, k3y, -is, ic, ha, hwb, hhb);
    if (!with_origin_terms) return hha * (l0 + l2) + hwa * (l1 + l3);  // origin = A's own centre
    float xu = pax * s - pay * c;  // cross(pa, u^)
    float xv = pax * c + pay * s;  // cross(pa, v^)
    return (l0 * (hha - xu) + l2 * (hha + xu)) + (l1 * (hwa - xv) + l3 * (hwa + xv));
}

// Double-precision twins of clip_len / edges_inside.  The fp32 boundary integral loses ~6e-8 * extent / |sin(delta)| of
// the area: harmless at the |delta| >= 1.2e-3 the rotated jitter normally leaves, but the reference's two jitter steps
// can also cancel (delta = 4 eps + eps' before the jitter => ~4e-6 after it; found by the 2 M-pair soak test on 180-degree
// boxes: 2e-2 IoU error).  Below kNearParallel the integral is evaluated in double; also used by the jitter-free naive
// RBFoV stage.
constexpr float kNearParallel = 2.5e-4f;
SPH_DEV double clip_len_d(double px, double py, double ux, double uy, double len, double hx, double hy) {
    const double iux = 1.0 / ux, iuy = 1.0 / uy;
    const double ax = (-hx - px) * iux, bx = (hx - px) * iux, ay = (-hy - py) * iuy, by = (hy - py) * iuy;
    const double lo = fmax(fmax(fmin(ax, bx), fmin(ay, by)), 0.0), hi = fmin(fmin(fmax(ax, bx), fmax(ay, by)), len);
    return fmax(hi - lo, 0.0);
}
SPH_DEV double edges_inside_d(double pax, double pay, double c, double s, double hwa, double hha, double hwb, double hhb,
                              bool with_origin_terms) {
    const double ux = hwa * c, uy = hwa * s, vx = -hha * s, vy = hha * c;
    const double l0 = clip_len_d(pax + ux + vx, pay + uy + vy, -c, -s, 2.0 * hwa, hwb, hhb);
    const double l1 = clip_len_d(pax - ux + vx, pay - uy + vy, s, -c, 2.0 * hha, hwb, hhb);
    const double l2 = clip_len_d(pax - ux - vx, pay - uy - vy, c, s, 2.0 * hwa, hwb, hhb);
    const double l3 = clip_len_d(pax + ux - vx, pay + uy - vy, -s, c, 2.0 * hha, hwb, hhb);
    if (!with_origin_terms) return hha * (l0 + l2) + hwa * (l1 + l3);
    const double xu = pax * s - pay * c, xv = pax * c + pay * s;
    return (l0 * (hha - xu) + l2 * (hha + xu)) + (l1 * (hwa - xv) + l3 * (hwa + xv));
}

// Twice the intersection area of two nearly parallel rectangles, in double and from ONE consistent description: A's
// centre (pax, pay) and rotation (c, s) in B's frame; (c, s) is re-normalised and B's centre in A's frame is derived
// from them.  The two passes of the boundary integral must agree on where nearly coincident edges cross: with
// positions / rotations rounded independently in fp32 (1e-7) the crossing moves by 1e-7 / |s| along the edge, i.e.
// by a few percent of the edge at |s| ~ 4e-6, and the two halves no longer add up to the area.
SPH_DEV double near_parallel_area2(double pax, double pay, double c, double s, double hwa, double hha, double hwb,
                                   double hhb) {
    const double n = 1.0 / sqrt(c * c + s * s);
    c *= n;
    s *= n;
    // pass 0: A's edges in B's frame (origin terms about B's centre); pass 1: B's edges in A's frame (rotation -s,
    // centre -R^T pa), integrated about B's own centre.  A rolled loop over the 8 edges keeps the register footprint
    // of this rare branch below that of the fp32 fast path around it.
    double total = 0.0;
#pragma nounroll
    for (int e = 0; e < 8; e++) {
        const bool second = e >= 4;
        const double px = second ? -(c * pax + s * pay) : pax, py = second ? -(-s * pax + c * pay) : pay;
        const double cc = c, ss = second ? -s : s;
        const double hw = second ? hwb : hwa, hh = second ? hhb : hha;   // the rectangle whose edges are walked
        const double hx = second ? hwa : hwb, hy = second ? hha : hhb;   // the rectangle they are clipped to
        const int k = e & 3;
        // corner k (+u+v, -u+v, -u-v, +u-v) and direction of the edge leaving it (-u, -v, +u, +v)
        const double su = (k == 0 || k == 3) ? 1.0 : -1.0, sv = (k <= 1) ? 1.0 : -1.0;
        const double kx = px + su * hw * cc - sv * hh * ss, ky = py + su * hw * ss + sv * hh * cc;
        const bool along_u = (k & 1) == 0;
        const double sgn = (k == 0 || k == 1) ? -1.0 : 1.0;
        const double ux = along_u ? sgn * cc : -sgn * ss, uy = along_u ? sgn * ss : sgn * cc;
        const double len = along_u ? 2.0 * hw : 2.0 * hh;
        const double l = clip_len_d(kx, ky, ux, uy, len, hx, hy);
        // distance of the edge's supporting line from the integration origin (B's centre)
        double dist = along_u ? hh : hw;
        if (!second) {
            const double xu = px * ss - py * cc, xv = px * cc + py * ss;
            dist += along_u ? (k == 0 ? -xu : xu) : (k == 1 ? -xv : xv);
        }
        total += l * dist;
    }
    return total;
}

// Area of the intersection of two rotated rectangles (x, y, w, h, a).
SPH_DEV float rect_intersection(const PBox& A, const PBox& B) {
    float sa = sinf(A.a), ca = cosf(A.a), sb = sinf(B.a), cb = cosf(B.a);
    float dx = B.x - A.x, dy = B.y - A.y;
    float c = ca * cb + sa * sb;  // cos(aA - aB)
    float s = sa * cb - ca * sb;  // sin(aA - aB)
    float ic = 1.0f / c, is = 1.0f / s;
    float hwa = 0.5f * A.w, hha = 0.5f * A.h, hwb = 0.5f * B.w, hhb = 0.5f * B.h;
    // A's centre in B's frame; B's centre in A's frame
    float pax = -(dx * cb + dy * sb), pay = -(dy * cb - dx * sb);
    float pbx = dx * ca + dy * sa, pby = dy * ca - dx * sa;
    // A's edges inside B, integrated about B's centre; B's edges inside A, same origin (= B's own centre)
    if (fabsf(s) < kNearParallel && s != 0.0f) {  // rare: see kNearParallel
        const double cd = (double)ca * cb + (double)sa * sb, sd = (double)sa * cb - (double)ca * sb;
        return 0.5f * fmaxf((float)near_parallel_area2(pax, pay, cd, sd, hwa, hha, hwb, hhb), 0.0f);
    }
    float t = edges_inside(pax, pay, c, s, ic, is, hwa, hha, hwb, hhb, A.w, A.h, true) +
              edges_inside(pbx, pby, c, -s, ic, -is, hwb, hhb, hwa, hha, B.w, B.h, false);
    return 0.5f * fmaxf(t, 0.0f);
}

SPH_DEV float planar_iou(const PBox& A, const PBox& B, int mode) {
    float inter = rect_intersection(A, B);
    float a1 = A.w * A.h, a2 = B.w * B.h;
    float base = mode == MODE_IOU ? (a1 + a2 - inter) : a1;
    return inter / base;
}

// Sph-IoU / FoV-IoU closed forms (sphdet/iou/approximate_ious.py:3-54) on spherically jittered boxes: the two cheap
// approximate backends of SphOverlaps2D (sph_iou_api.py:128-175).  ~40 VALU per pair: the one genuinely HBM-bound
// operator of the package.
template <bool FOV>
SPH_DEV float approx_iou(const float (&g_)[5], const float (&p_)[5]) {
    float g0 = g_[0], p0 = p_[0];
    if (fabsf(g0 - p0) > 180.0f) {  // standardize_spherical_box :59-79
        g0 = fmodf(g0 + 180.0f, 360.0f);
        p0 = fmodf(p0 + 180.0f, 360.0f);
    }
    const float hpi = (float)(3.141592653589793 / 2);
    float thg = g0 * kDeg2Rad - kPi, phg = hpi - g_[1] * kDeg2Rad;  // angle2radian 'convention' :81-99
    float thp = p0 * kDeg2Rad - kPi, php = hpi - p_[1] * kDeg2Rad;
    float ag = g_[2] * kDeg2Rad, bg = g_[3] * kDeg2Rad, ap = p_[2] * kDeg2Rad, bp = p_[3] * kDeg2Rad;
    float ag2 = ag / 2.0f, bg2 = bg / 2.0f, ap2 = ap / 2.0f, bp2 = bp / 2.0f;
    float tmin, tmax;
    if (FOV) {
        float delta = (thp - thg) * cosf((phg + php) / 2.0f);
        tmin = fmaxf(-ag2, delta - ap2);
        tmax = fminf(ag2, delta + ap2);
    } else {
        tmin = fmaxf(thg - ag2, thp - ap2);
        tmax = fminf(thg + ag2, thp + ap2);
    }
    float pmin = fmaxf(phg - bg2, php - bp2), pmax = fminf(phg + bg2, php + bp2);
    float ai = fmaxf(tmax - tmin, 0.0f) * fmaxf(pmax - pmin, 0.0f);
    float au = ag * bg + ap * bp - ai;
    return ai / (au + 1e-8f);
}

// Naive-IoU (sph_iou_api.py:179-197): boxes mapped to ERP pixels by Sph2PlanarBoxTransform('sph2pix') with the default
// img_size (512, 1024) (box_formator.py:76-83, :161-178), then mmcv.ops.bbox_overlaps on xyxy (BFoV) or
// mmcv.ops.box_iou_rotated on (x, y, w, h, -deg2rad(gamma)) (RBFoV).  No jitter and no clamp in the reference.
// The planar stage has no jitter in front of it here, so exactly parallel rectangles (equal gamma: every
// BFoV-like detection) are intersected as axis-aligned boxes in their common frame instead of through the
// edge-crossing integral, which needs transversal edges.
// `tan_form`: Sph2PlanarBoxTransform('sph2tan') (box_formator.py:98-106): w = 2R tan(alpha / 2), h = 2R tan(beta / 2) with
// 2R = img_w / pi, instead of the 'sph2pix' proportions (:76-83)
template <int DIM>
SPH_DEV float naive_iou(const float (&b1)[5], const float (&b2)[5], bool tan_form = false) {
    const float W = 1024.0f, H = 512.0f;
    const float xa = (b1[0] / 360.0f) * W, ya = (b1[1] / 180.0f) * H, xb = (b2[0] / 360.0f) * W, yb = (b2[1] / 180.0f) * H;
    float wa, ha, wb, hb;
    if (tan_form) {
        const float twoR = (float)(1024.0 / 3.141592653589793);
        wa = twoR * tanf((b1[2] * kDeg2Rad) / 2.0f); ha = twoR * tanf((b1[3] * kDeg2Rad) / 2.0f);
        wb = twoR * tanf((b2[2] * kDeg2Rad) / 2.0f); hb = twoR * tanf((b2[3] * kDeg2Rad) / 2.0f);
    } else {
        wa = (b1[2] / 360.0f) * W; ha = (b1[3] / 180.0f) * H; wb = (b2[2] / 360.0f) * W; hb = (b2[3] / 180.0f) * H;
    }
    if (DIM == 4) {  // xywh2xyxy (box_formator.py:25-31) + bbox_overlaps(mode='iou', aligned, offset=0)
        const float ax1 = xa - wa / 2.0f, ay1 = ya - ha / 2.0f, ax2 = xa + wa / 2.0f, ay2 = ya + ha / 2.0f;
        const float bx1 = xb - wb / 2.0f, by1 = yb - hb / 2.0f, bx2 = xb + wb / 2.0f, by2 = yb + hb / 2.0f;
        const float area_a = (ax2 - ax1) * (ay2 - ay1), area_b = (bx2 - bx1) * (by2 - by1);
        const float iw = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.0f), ih = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.0f);
        const float inter = iw * ih;
        return inter / fmaxf(area_a + area_b - inter, 0.0f);
    }
    // Rotated boxes without a jitter in front: near-parallel edges (equal-ish gamma is the common case) make the
    // edge-crossing integral ill-conditioned in fp32 (1 / sin(delta) amplification), so this stage runs in double.
    const double aa = -(double)(b1[4] * kDeg2Rad), ab = -(double)(b2[4] * kDeg2Rad);
    const double sa = sin(aa), ca = cos(aa), sb = sin(ab), cb = cos(ab);
    const double c = ca * cb + sa * sb, s = sa * cb - ca * sb;
    const double dx = (double)xb - (double)xa, dy = (double)yb - (double)ya;
    const double hwa = 0.5 * wa, hha = 0.5 * ha, hwb = 0.5 * wb, hhb = 0.5 * hb;
    const double pax = -(dx * cb + dy * sb), pay = -(dy * cb - dx * sb);
    double inter;
    if (fabs(s) < 1e-12 || fabs(c) < 1e-12) {  // parallel or perpendicular: axis-aligned in B's frame
        const bool par = fabs(s) < 1e-12;
        const double ex = par ? hwa : hha, ey = par ? hha : hwa;
        const double ox = fmax(fmin(pax + ex, hwb) - fmax(pax - ex, -hwb), 0.0);
        const double oy = fmax(fmin(pay + ey, hhb) - fmax(pay - ey, -hhb), 0.0);
        inter = ox * oy;
    } else {
        const double pbx = dx * ca + dy * sa, pby = dy * ca - dx * sa;
        const double t = edges_inside_d(pax, pay, c, s, hwa, hha, hwb, hhb, true) +
                         edges_inside_d(pbx, pby, c, -s, hwb, hhb, hwa, hha, false);
        inter = 0.5 * fmax(t, 0.0);
    }
    const double area = (double)wa * ha + (double)wb * hb;
    return (float)(inter / (area - inter));
}

// _sph2pob_iou_auxiliary for one pair — sph_iou_api.py:48-86
template <int VARIANT, int DIM>
SPH_DEV float pair_iou(const float (&in1)[5], const float (&in2)[5], int mode, int edge, int angle) {
    if (pair_has_nan<DIM>(in1, in2)) return __builtin_nanf("");   // torch.clamp propagates NaN (sph_iou_api.py:86)
    float b1[5], b2[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { b1[k] = in1[k]; b2[k] = in2[k]; }
    jitter_spherical<DIM>(b1, b2);
    if (VARIANT == VARIANT_SPH_IOU || VARIANT == VARIANT_FOV_IOU)
        return fminf(fmaxf(approx_iou<VARIANT == VARIANT_FOV_IOU>(b1, b2), 0.0f), 1.0f);
    PBox p1, p2;
    transform<VARIANT, DIM>(b1, b2, edge, angle, p1, p2);
    jitter_rotated(p1, p2);
    // legacy only: asin(sqrt(q < 0)) is NaN in the reference too; mmcv's kernel then finds no intersection
    // point (every comparison with NaN is false) and returns 0.
    if (VARIANT == VARIANT_LEGACY && !(p2.x == p2.x)) return 0.0f;
    float iou = planar_iou(p1, p2, mode);
    return fminf(fmaxf(iou, 0.0f), 1.0f);
}

}  // namespace sph2pob

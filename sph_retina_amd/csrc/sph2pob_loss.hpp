// Sph2PobIoULoss forward + hand-derived backward for one pair (device code, gfx950).
//
// Forward restates  Sph2PobTransfrom.new_forward (sphdet/losses/sph2pob_transform.py:24-35: clone, spherical
// jitter, sph2pob_standard(...,'rad'), rotated jitter)  followed by  obb_iou_loss
// (sphdet/losses/sph2pob_iou_loss.py:104-196: IoU | GIoU | DIoU | CIoU with obb2hbb_xyxy,
// sphdet/bbox/box_formator.py:33-54).  The reference's backward is torch autograd through ~150 recorded ops
// (transform + mmcv diff_iou_rotated_2d + penalty terms); here it is one closed-form adjoint:
//
//   planar boxes as functions of the inputs (normal branch of compute_rotate_matrix_auto):
//       pred   P = (-A/2, pi/2, k*alpha_g, k*beta_g, B_g - k*gamma_g)      k = pi/180
//       target T = (+A/2, pi/2, k*alpha_p, k*beta_p, B_p - k*gamma_p)
//       A   = angle(c_g, c_p)                      (great-circle distance of the centres)
//       B_g = atan2( c_p.d_g, -c_p.e_g)            (bearing of p seen from g in g's (south, east) frame)
//       B_p = atan2(-c_g.d_p,  c_g.e_p)
//   dI of the rectangle intersection = boundary transport: each clipped edge piece [s0, s1] of box X moves the
//   area by  len * (n . dc)  under translation, len/2 under growth of the matching extent and
//   -(s1^2 - s0^2)/2 under rotation about the centre (s measured from the edge midpoint, counter-clockwise).
//   The acos(clamp(.)) of the reference has zero gradient where the clamp is active (A/2 < 4.88e-4,
//   |cos a| > 1 - 1e-7); in-place clamps of the two jitters gate the gradient the same way torch.clamp_ does.
#pragma once
#include "sph2pob_device.hpp"

namespace sph2pob {

enum : int { LOSS_IOU = 0, LOSS_GIOU = 1, LOSS_DIOU = 2, LOSS_CIOU = 3 };

struct ClipIv { float lo, hi; };
// clipped interval of P(tau) = p + tau*u, tau in [0, len], inside |x| <= hx, |y| <= hy (see clip_len)
SPH_DEV ClipIv clip_iv(float px, float py, float iux, float iuy, float len, float hx, float hy) {
    float ax = (-hx - px) * iux, bx = (hx - px) * iux;
    float ay = (-hy - py) * iuy, by = (hy - py) * iuy;
    float lo = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), 0.0f);
    float hi = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), len);
    // empty interval (also lo = +inf for an edge parallel to and outside a slab): collapse to [0, 0] so that the
    // length and the rotation term (hi + lo - len) * (hi - lo) stay finite
    const bool empty = !(hi > lo);
    return ClipIv{empty ? 0.0f : lo, empty ? 0.0f : hi};
}

// intersection area + its gradient w.r.t. one rectangle's own (centre-along-its-axes, w, h, a)
struct EdgeSet {
    float area2;      // twice the boundary-integral contribution (see edges_inside)
    float dcu, dcv;   // dI/d(centre) along the rectangle's own u and v axes
    float dw, dh, da; // dI/dw, dI/dh, dI/da
};
SPH_DEV EdgeSet edges_inside_grad(float pax, float pay, float c, float s, float ic, float is, float hwa, float hha,
                                  float hwb, float hhb, float wa, float ha, bool with_origin_terms) {
    float ux = hwa * c, uy = hwa * s, vx = -hha * s, vy = hha * c;
    float k0x = pax + ux + vx, k0y = pay + uy + vy;
    float k1x = pax - ux + vx, k1y = pay - uy + vy;
    float k2x = pax - ux - vx, k2y = pay - uy - vy;
    float k3x = pax + ux - vx, k3y = pay + uy - vy;
    ClipIv i0 = clip_iv(k0x, k0y, -ic, -is, wa, hwb, hhb);  // dir -u, outward normal +v
    ClipIv i1 = clip_iv(k1x, k1y, is, -ic, ha, hwb, hhb);   // dir -v, outward normal -u
    ClipIv i2 = clip_iv(k2x, k2y, ic, is, wa, hwb, hhb);    // dir +u, outward normal -v
    ClipIv i3 = clip_iv(k3x, k3y, -is, ic, ha, hwb, hhb);   // dir +v, outward normal +u
    float l0 = i0.hi - i0.lo, l1 = i1.hi - i1.lo, l2 = i2.hi - i2.lo, l3 = i3.hi - i3.lo;
    EdgeSet e;
    if (with_origin_terms) {
        float xu = pax * s - pay * c, xv = pax * c + pay * s;
        e.area2 = (l0 * (hha - xu) + l2 * (hha + xu)) + (l1 * (hwa - xv) + l3 * (hwa + xv));
    } else {
        e.area2 = hha * (l0 + l2) + hwa * (l1 + l3);
    }
    e.dcu = l3 - l1;
    e.dcv = l0 - l2;
    e.dw = 0.5f * (l1 + l3);
    e.dh = 0.5f * (l0 + l2);
    // rotation: -(s1^2 - s0^2)/2 with s = tau - len/2  ==  -(hi - lo) * ((hi + lo) - len) / 2
    e.da = -0.5f * ((l0 * ((i0.hi + i0.lo) - wa) + l2 * ((i2.hi + i2.lo) - wa)) +
                    (l1 * ((i1.hi + i1.lo) - ha) + l3 * ((i3.hi + i3.lo) - ha)));
    return e;
}

struct PlanarGrad { float x, w, h, a; };  // d(.)/d(x, w, h, a) of one planar box (y never carries gradient)

// What the front end (reference-order or closed-form) hands to the shared loss core: the two planar boxes after the
// rotated jitter in the frame "pred box at the origin" (target centre = (dx, dy)), gradient gates of every clamp /
// acos floor on the way, and the trig of the jittered spherical inputs for the chain rule.
struct LossFront {
    float dx, dy, ca, sa, cb, sb, wg, hg, wp, hp, c, s;   // (c, s) = cos / sin of (a_P - a_T)
    bool g_A, g_ag, g_ap, g_wg, g_hg, g_wp, g_hp;
    float sg, cg, sp, cp, sD, cD;  // sin/cos(phi_g), sin/cos(phi_p), sin/cos(theta_p - theta_g)
};

// reference-order front end: sph2pob_standard in the reference's fp32 operation order
template <int DIM>
SPH_DEV void loss_front_reference(const float (&b1)[5], const float (&b2)[5], LossFront& f) {
    PBox P, T;
    transform_standard<DIM>(b1, b2, EDGE_ARC, ANGLE_EQUATOR, P, T);
    const PBox P0 = P, T0 = T;  // before the rotated jitter: its clamp gates look at these
    jitter_rotated(P, T);
    f.dx = T.x - P.x; f.dy = T.y - P.y;
    f.sa = sinf(P.a); f.ca = cosf(P.a); f.sb = sinf(T.a); f.cb = cosf(T.a);
    f.c = f.ca * f.cb + f.sa * f.sb; f.s = f.sa * f.cb - f.ca * f.sb;
    f.wg = P.w; f.hg = P.h; f.wp = T.w; f.hp = T.h;
    {   // rotated-jitter clamp gates (sph_iou_api.py:237-240: in-place clamp_ => zero gradient outside)
        const float e = (float)kEpsS, e2 = (float)(2 * kEpsS), e5 = (float)(5 * kEpsS);
        const float ea = (float)kEpsA, ea2 = (float)(2 * kEpsA);
        bool similar = (fabsf(P0.x - T0.x) < e) | (fabsf(P0.w - T0.w) < e) | (fabsf(P0.h - T0.h) < e) |
                       (fabsf(P0.a - T0.a) < e);
        float pw = P0.w + (similar ? e2 : 0.0f), ph = P0.h + (similar ? e2 : 0.0f), pa = P0.a + (similar ? e : 0.0f);
        float tw = T0.w + (similar ? e : 0.0f), th = T0.h + (similar ? e : 0.0f), ta = T0.a + (similar ? e5 : 0.0f);
        if (fabsf(pa - ta) < ea) { pa += ea; ta += ea2; }
        const double pi = 3.141592653589793;
        f.g_wg = pw >= (float)(2 * kEpsA / 10); f.g_hg = ph >= (float)(2 * kEpsA / 10);
        f.g_wp = tw >= (float)(kEpsA / 10);     f.g_hp = th >= (float)(kEpsA / 10);
        bool in_a = !(pa < (float)(-2 * pi + 2 * kEpsA) || pa > (float)(2 * pi - kEpsA));
        bool in_b = !(ta < (float)(-2 * pi + kEpsA) || ta > (float)(2 * pi - 2 * kEpsA));
        // acos(clamp(., -1+1e-7, 1-1e-7)) of compute_internal_angle: zero gradient where the clamp is active
        f.g_ag = in_a && fabsf(cosf(P0.a)) < kClampHi;
        f.g_ap = in_b && fabsf(cosf(T0.a)) < kClampHi;
    }
    SBox g = load_sbox<DIM>(b1), p = load_sbox<DIM>(b2);
    f.sg = g.sp; f.cg = g.cp; f.sp = p.sp; f.cp = p.cp;
    f.cD = p.ct * g.ct + p.st * g.st;
    f.sD = p.st * g.ct - p.ct * g.st;
    // x' = +-acos(clamp(cos(A/2))) of compute_spherical_coordinate: gate on A/2 >= acos(1 - 1e-7)
    float N = f.sp * f.cg * f.cD - f.cp * f.sg, D = -f.sp * f.sD;
    float C = f.cg * f.cp + f.sg * f.sp * f.cD;
    f.g_A = atan2f(sqrtf(N * N + D * D), C) > 2.0f * 4.8828125e-4f;
}

}  // namespace sph2pob
#include "sph2pob_fast.hpp"

namespace sph2pob {
// 1 / b and a / b from the hardware reciprocal plus one Newton step (~1 ulp).  The IEEE divide expands to ~12
// instructions, 5 of them at the slow issue rate (v_div_scale x2, v_div_fmas, v_div_fixup, v_rcp): the loss kernels
// are VALU-issue-bound and divided 7 (forward) to 19 (backward) times per pair.  Only for denominators that cannot be
// 0 / inf (sums with eps, squares of clamped extents); the clip reciprocals use the bare, clamped v_rcp.
SPH_DEV float rcp_nr(float b) {
    float r = fast_rcp(b);
    return r * (2.0f - b * r);
}
SPH_DEV float fdiv(float a, float b) { return a * rcp_nr(b); }
}  // namespace sph2pob
namespace sph2pob {

// closed-form front end (sph2pob_fast.hpp): same planar boxes, ~4x fewer instructions
template <int DIM, int VARIANT = VARIANT_STANDARD>
SPH_DEV void loss_front_fast(const float (&b1)[5], const float (&b2)[5], LossFront& f, int edge = EDGE_ARC,
                             bool rot_jitter = true) {
    FastTrig t;
    PlanarPair q;
    // rot_jitter == "the caller ran jitter_spherical first", i.e. the angles are clamped into the range of the cheap trig
    if (rot_jitter) lean_front<VARIANT, DIM, true, true>(b1, b2, edge, q, &t);
    else lean_front<VARIANT, DIM, true, false>(b1, b2, edge, q, &t);
    if (!rot_jitter) { q.g_wg = q.g_hg = q.g_wp = q.g_hp = true; }
    f.dx = q.dx; f.dy = q.dy; f.ca = q.ca; f.sa = q.sa; f.cb = q.cb; f.sb = q.sb; f.c = q.c; f.s = q.s;
    f.wg = q.wg; f.hg = q.hg; f.wp = q.wp; f.hp = q.hp;
    f.g_A = q.g_A; f.g_ag = q.g_ag; f.g_ap = q.g_ap; f.g_wg = q.g_wg; f.g_hg = q.g_hg; f.g_wp = q.g_wp; f.g_hp = q.g_hp;
    f.sg = t.sg; f.cg = t.cg; f.sp = t.sp; f.cp = t.cp; f.sD = t.sD; f.cD = t.cD;
}

// Chain rule from gradients w.r.t. the two planar boxes (x, w, h, a) to gradients w.r.t. the spherical inputs
// (degrees).  x_split: P.x = -x_split*A and T.x = +(1 - x_split)*A ... expressed as dT.x/dA - dP.x/dA = 1:
// standard (x = -+A/2) passes 0.5, efficient (P.x = 0, T.x = A) passes 0.  Gates: acos(clamp) floors and the rotated
// jitter's clamps (from the front end), the spherical jitter's in-place clamps (here, when `jitter`).
template <int DIM>
SPH_DEV void planar_to_spherical_grads(const LossFront& f, PlanarGrad gP, PlanarGrad gT, float x_split,
                                       const float (&pred)[5], const float (&target)[5], bool jitter, int edge,
                                       float (&gpred)[5], float (&gtarget)[5]) {
    if (!f.g_wg) gP.w = 0.0f;
    if (!f.g_hg) gP.h = 0.0f;
    if (!f.g_wp) gT.w = 0.0f;
    if (!f.g_hp) gT.h = 0.0f;
    float GA = f.g_A ? ((1.0f - x_split) * gT.x - x_split * gP.x) : 0.0f;
    float Gag = f.g_ag ? gP.a : 0.0f;
    float Gap = f.g_ap ? gT.a : 0.0f;
    const float sg = f.sg, cg = f.cg, sp = f.sp, cp = f.cp, sD = f.sD, cD = f.cD;
    float N = sp * cg * cD - cp * sg;    //  c_p . d_g
    float D = -sp * sD;                  // -c_p . e_g
    float Np = cg * sp - sg * cp * cD;   // -c_g . d_p
    float Dp = -sg * sD;                 //  c_g . e_p
    float C = cg * cp + sg * sp * cD;    //  cos A
    float sin2 = N * N + D * D;          //  sin^2 A
    float inv_s = fast_rsq(sin2);
    inv_s = sin2 > 1e-20f ? inv_s * (1.5f - 0.5f * sin2 * inv_s * inv_s) : 0.0f;  // 1 / sin A, one Newton step
    float inv_s2 = inv_s * inv_s;
    // dA = -dC / sinA
    float dA_phg = -N * inv_s, dA_php = Np * inv_s;
    float dA_thg = -(sg * sp * sD) * inv_s, dA_thp = (sg * sp * sD) * inv_s;
    // dB_g = (D dN - N dD) / sin^2 A
    float dBg_phg = (D * (-C)) * inv_s2;
    float dBg_thg = (D * (sp * cg * sD) - N * (sp * cD)) * inv_s2;
    float dpdg = cp * cg * cD + sp * sg;  // d_p . d_g
    float dBg_php = (D * dpdg - N * (-cp * sD)) * inv_s2;
    float dBg_thp = (D * (-sp * cg * sD) - N * (-sp * cD)) * inv_s2;
    // dB_p = (D' dN' - N' dD') / sin^2 A
    float dBp_php = (Dp * C) * inv_s2;
    float dBp_thp = (Dp * (sg * cp * sD) - Np * (-sg * cD)) * inv_s2;
    float dBp_phg = (Dp * (-dpdg) - Np * (-cg * sD)) * inv_s2;
    float dBp_thg = (Dp * (-sg * cp * sD) - Np * (sg * cD)) * inv_s2;

    float r_g[5], r_p[5];  // gradients w.r.t. radians
    r_g[0] = GA * dA_thg + Gag * dBg_thg + Gap * dBp_thg;
    r_g[1] = GA * dA_phg + Gag * dBg_phg + Gap * dBp_phg;
    r_g[2] = gP.w; r_g[3] = gP.h; r_g[4] = -Gag;
    r_p[0] = GA * dA_thp + Gag * dBg_thp + Gap * dBp_thp;
    r_p[1] = GA * dA_php + Gag * dBg_php + Gap * dBp_php;
    r_p[2] = gT.w; r_p[3] = gT.h; r_p[4] = -Gap;

    // spherical-jitter shifts / clamps (sph_iou_api.py:244-258): gates look at the pre-clamp values
    const float eps1 = (float)kEpsS, eps2 = (float)(2 * kEpsS);
    bool similar = false;
    if (jitter) {
#pragma unroll
        for (int k = 0; k < DIM; k++) similar |= fabsf(pred[k] - target[k]) < eps1;
    }
#pragma unroll
    for (int k = 0; k < 5; k++) {
        if (k >= DIM) { gpred[k] = 0.0f; gtarget[k] = 0.0f; continue; }
        float x1 = pred[k] - (similar ? eps2 : 0.0f), x2 = target[k] + (similar ? eps1 : 0.0f);
        float hi1 = k == 0 ? (float)(360.0 - kEpsS) : (float)(180.0 - kEpsS);
        float hi2 = k == 0 ? (float)(360.0 - 2 * kEpsS) : (float)(180.0 - 2 * kEpsS);
        bool in1 = !jitter || k == 4 || (x1 >= eps2 && x1 <= hi1);
        bool in2 = !jitter || (k == 4 ? (x2 >= (float)(-360.0 + 2 * kEpsS) && x2 <= (float)(360.0 - 2 * kEpsS))
                                      : (x2 >= eps1 && x2 <= hi2));
        float s1 = kDeg2Rad, s2 = kDeg2Rad;
        if ((k == 2 || k == 3) && edge != EDGE_ARC) {  // d(edge length)/d(fov): chord cos(f/2), tangent 1/cos^2(f/2)
            float jg = in1 ? fminf(fmaxf(x1, jitter ? eps2 : x1), jitter ? hi1 : x1) : x1;
            float jp = in2 ? fminf(fmaxf(x2, jitter ? eps1 : x2), jitter ? hi2 : x2) : x2;
            float c1 = cosf(0.5f * jg * kDeg2Rad), c2 = cosf(0.5f * jp * kDeg2Rad);
            s1 *= edge == EDGE_CHORD ? c1 : 1.0f / (c1 * c1);
            s2 *= edge == EDGE_CHORD ? c2 : 1.0f / (c2 * c2);
        }
        gpred[k] = in1 ? r_g[k] * s1 : 0.0f;
        gtarget[k] = in2 ? r_p[k] * s2 : 0.0f;
    }
}

// Per-pair loss element; when BWD, also d(loss)/d(pred[0..DIM)) and d(loss)/d(target[0..DIM)) in 1/degree.
template <int DIM, bool BWD, bool FAST>
SPH_DEV float pair_loss(const float (&pred)[5], const float (&target)[5], int loss_mode, float eps, float* iou_out,
                        float (&gpred)[5], float (&gtarget)[5]) {
    if (pair_has_nan<DIM>(pred, target)) {
        // torch propagates a NaN coordinate into the loss element and, through autograd, into every gradient of the
        // pair; the clamps / min / max on the way here would drop it and train on garbage
        const float qnan = __builtin_nanf("");
        if (iou_out) *iou_out = qnan;
#pragma unroll
        for (int k = 0; k < 5; k++) { gpred[k] = k < DIM ? qnan : 0.0f; gtarget[k] = k < DIM ? qnan : 0.0f; }
        return qnan;
    }
    float b1[5], b2[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { b1[k] = pred[k]; b2[k] = target[k]; }
    jitter_spherical<DIM>(b1, b2);
    LossFront f;
    if (FAST) loss_front_fast<DIM>(b1, b2, f);
    else loss_front_reference<DIM>(b1, b2, f);
    const float sa = f.sa, ca = f.ca, sb = f.sb, cb = f.cb, dx = f.dx, dy = f.dy;
    struct { float x, y, w, h; } P{0.0f, 0.0f, f.wg, f.hg}, T{dx, dy, f.wp, f.hp};

    // ---- planar IoU (value of mmcv diff_iou_rotated_2d: sphdet/iou/diff_iou_rotated.py:325-343) ----
    const float c = f.c, s = f.s;
    // clamped reciprocals: exactly parallel edges (s == 0 after the jitter bumps) give finite, correctly ordered bounds
    float ic = fminf(fmaxf(fast_rcp(c), -1e18f), 1e18f), is = fminf(fmaxf(fast_rcp(s), -1e18f), 1e18f);
    float hwa = 0.5f * P.w, hha = 0.5f * P.h, hwb = 0.5f * T.w, hhb = 0.5f * T.h;
    float pax = -(dx * cb + dy * sb), pay = -(dy * cb - dx * sb);
    float pbx = dx * ca + dy * sa, pby = dy * ca - dx * sa;
    EdgeSet eA = edges_inside_grad(pax, pay, c, s, ic, is, hwa, hha, hwb, hhb, P.w, P.h, true);
    EdgeSet eB = edges_inside_grad(pbx, pby, c, -s, ic, -is, hwb, hhb, hwa, hha, T.w, T.h, false);
    float I = 0.5f * fmaxf(eA.area2 + eB.area2, 0.0f);
    {   // near-parallel planar boxes (the two jitter steps cancelled: DESIGN.md §9): the value of the intersection comes
        // from the first-order form; the boundary-transport gradient terms above stay (their relative error there is the
        // integral's, a few percent at worst, on ~1e-4 of the pairs).  Wave-uniform guard: the common wave skips the code.
        const bool near = fminf(fabsf(s), fabsf(c)) < kNearParallel;
#if defined(__HIP_DEVICE_COMPILE__)
        if (__ballot(near) != 0ull)
#endif
            if (near) I = fmaxf(near_parallel_inter(pax, pay, c, s, hwa, hha, hwb, hhb), 0.0f);
    }
    float S1 = P.w * P.h, S2 = T.w * T.h;
    float U = S1 + S2 - I;
    float iou_raw = fdiv(I, U);
    float iou = fminf(fmaxf(iou_raw, 0.0f), 1.0f);
    if (iou_out) *iou_out = iou;

    // ---- penalty terms (sph2pob_iou_loss.py:139-194) ----
    float aca = fabsf(ca), asa = fabsf(sa), acb = fabsf(cb), asb = fabsf(sb);
    float Wg = aca * P.w + asa * P.h, Hg = asa * P.w + aca * P.h;
    float Wp = acb * T.w + asb * T.h, Hp = asb * T.w + acb * T.h;
    float x1g = P.x - Wg / 2.0f, x2g = P.x + Wg / 2.0f, y1g = P.y - Hg / 2.0f, y2g = P.y + Hg / 2.0f;
    float x1p = T.x - Wp / 2.0f, x2p = T.x + Wp / 2.0f, y1p = T.y - Hp / 2.0f, y2p = T.y + Hp / 2.0f;
    float cw = fmaxf(fmaxf(x2g, x2p) - fminf(x1g, x1p), 0.0f);
    float ch = fmaxf(fmaxf(y2g, y2p) - fminf(y1g, y1p), 0.0f);
    float loss, pen_ratio = 0.0f, c2 = 0.0f, rho2 = 0.0f, v = 0.0f, alpha = 0.0f, dv = 0.0f;
    float iw = 0.0f, ih = 0.0f, ae = 0.0f, au = 0.0f;
    if (loss_mode == LOSS_IOU) {
        loss = 1.0f - iou;
    } else if (loss_mode == LOSS_GIOU) {
        iw = fmaxf(fminf(x2g, x2p) - fmaxf(x1g, x1p), 0.0f);
        ih = fmaxf(fminf(y2g, y2p) - fmaxf(y1g, y1p), 0.0f);
        ae = cw * ch;
        au = S1 + S2 - iw * ih;
        pen_ratio = fdiv(ae - au, ae + eps);
        loss = 1.0f - (iou - fminf(fmaxf(pen_ratio, 0.0f), 1.0f));
    } else {
        c2 = cw * cw + ch * ch + eps;
        rho2 = dx * dx + dy * dy;
        pen_ratio = fdiv(rho2, c2);
        float pen = fminf(fmaxf(pen_ratio, 0.0f), 1.0f);
        if (loss_mode == LOSS_CIOU) {
            const float factor = (float)(4.0 / (3.141592653589793 * 3.141592653589793));
            dv = atan2_r(T.w, T.h + eps) - atan2_r(P.w, P.h + eps);   // widths and heights are positive
            v = factor * (dv * dv);
            alpha = fdiv((iou > 0.5f ? 1.0f : 0.0f) * v, 1.0f - iou + v + eps);
            pen = pen + alpha * v;
        }
        loss = 1.0f - (iou - pen);
    }
    if (!BWD) return loss;

    // =========================== adjoint ===========================
    PlanarGrad gP{0, 0, 0, 0}, gT{0, 0, 0, 0};
    // -- IoU term: L = 1 - iou  (torch.clamp passes gradient on the closed range [0, 1]) --
    if (iou_raw >= 0.0f && iou_raw <= 1.0f) {
        float inv_u = rcp_nr(U), inv_u2 = inv_u * inv_u;
        float LI = -(S1 + S2) * inv_u2;  // dL/dI
        float LS = I * inv_u2;           // dL/dS1 = dL/dS2
        if (eA.area2 + eB.area2 <= 0.0f) LI = 0.0f;
        // centre derivative in world x: dcu * cos(a) - dcv * sin(a)
        gP.x += LI * (eA.dcu * ca - eA.dcv * sa);
        gT.x += LI * (eB.dcu * cb - eB.dcv * sb);
        gP.w += LI * eA.dw + LS * P.h;  gP.h += LI * eA.dh + LS * P.w;  gP.a += LI * eA.da;
        gT.w += LI * eB.dw + LS * T.h;  gT.h += LI * eB.dh + LS * T.w;  gT.a += LI * eB.da;
    }
    if (loss_mode != LOSS_IOU) {
        // d(hbb extents)/d(w, h, a)
        float sgca = ca >= 0.0f ? 1.0f : -1.0f, sgsa = sa >= 0.0f ? 1.0f : -1.0f;
        float sgcb = cb >= 0.0f ? 1.0f : -1.0f, sgsb = sb >= 0.0f ? 1.0f : -1.0f;
        float dWg_a = -sgca * sa * P.w + sgsa * ca * P.h, dHg_a = sgsa * ca * P.w - sgca * sa * P.h;
        float dWp_a = -sgcb * sb * T.w + sgsb * cb * T.h, dHp_a = sgsb * cb * T.w - sgcb * sb * T.h;
        // which box supplies each side of the enclosing / intersecting hbb
        float g_x2 = x2g >= x2p ? 1.0f : 0.0f, g_x1 = x1g <= x1p ? 1.0f : 0.0f;
        float g_y2 = y2g >= y2p ? 1.0f : 0.0f, g_y1 = y1g <= y1p ? 1.0f : 0.0f;
        float Lcw = 0.0f, Lch = 0.0f;  // dL/dcw, dL/dch
        if (loss_mode == LOSS_GIOU) {
            if (pen_ratio >= 0.0f && pen_ratio <= 1.0f) {
                float den = ae + eps;
                float iden = rcp_nr(den), Rae = (eps + au) * (iden * iden), Rau = -iden;
                Lcw = Rae * ch; Lch = Rae * cw;
                // au = S1 + S2 - iw*ih
                gP.w += Rau * P.h; gP.h += Rau * P.w; gT.w += Rau * T.h; gT.h += Rau * T.w;
                float Liw = -Rau * ih, Lih = -Rau * iw;
                if (fminf(x2g, x2p) - fmaxf(x1g, x1p) < 0.0f) Liw = 0.0f;
                if (fminf(y2g, y2p) - fmaxf(y1g, y1p) < 0.0f) Lih = 0.0f;
                // iw = min(x2g, x2p) - max(x1g, x1p)
                float h_x2 = x2g <= x2p ? 1.0f : 0.0f, h_x1 = x1g >= x1p ? 1.0f : 0.0f;
                float h_y2 = y2g <= y2p ? 1.0f : 0.0f, h_y1 = y1g >= y1p ? 1.0f : 0.0f;
                float iwWg = 0.5f * (h_x2 + h_x1), iwWp = 0.5f * ((1 - h_x2) + (1 - h_x1));
                float ihHg = 0.5f * (h_y2 + h_y1), ihHp = 0.5f * ((1 - h_y2) + (1 - h_y1));
                gP.x += Liw * (h_x2 - h_x1);
                gT.x += Liw * ((1 - h_x2) - (1 - h_x1));
                float LWg = Liw * iwWg, LWp = Liw * iwWp, LHg = Lih * ihHg, LHp = Lih * ihHp;
                gP.w += LWg * aca + LHg * asa; gP.h += LWg * asa + LHg * aca; gP.a += LWg * dWg_a + LHg * dHg_a;
                gT.w += LWp * acb + LHp * asb; gT.h += LWp * asb + LHp * acb; gT.a += LWp * dWp_a + LHp * dHp_a;
            }
        } else {
            if (pen_ratio >= 0.0f && pen_ratio <= 1.0f) {
                float Lrho = rcp_nr(c2), Lc2 = -rho2 * (Lrho * Lrho);
                gT.x += Lrho * 2.0f * dx;
                gP.x -= Lrho * 2.0f * dx;
                Lcw = Lc2 * 2.0f * cw; Lch = Lc2 * 2.0f * ch;
            }
            if (loss_mode == LOSS_CIOU) {
                const float factor = (float)(4.0 / (3.141592653589793 * 3.141592653589793));
                float Ldv = alpha * factor * 2.0f * dv;  // alpha is a constant (torch.no_grad) :188-189
                float hpe = T.h + eps, hge = P.h + eps;
                float qp = rcp_nr(hpe * hpe + T.w * T.w), qg = rcp_nr(hge * hge + P.w * P.w);
                gT.w += Ldv * hpe * qp;  gT.h -= Ldv * T.w * qp;
                gP.w -= Ldv * hge * qg;  gP.h += Ldv * P.w * qg;
            }
        }
        if (fmaxf(x2g, x2p) - fminf(x1g, x1p) < 0.0f) Lcw = 0.0f;
        if (fmaxf(y2g, y2p) - fminf(y1g, y1p) < 0.0f) Lch = 0.0f;
        // cw = max(x2g, x2p) - min(x1g, x1p), x2 = x + W/2, x1 = x - W/2
        gP.x += Lcw * (g_x2 - g_x1);
        gT.x += Lcw * ((1 - g_x2) - (1 - g_x1));
        float LWg = Lcw * 0.5f * (g_x2 + g_x1), LWp = Lcw * 0.5f * ((1 - g_x2) + (1 - g_x1));
        float LHg = Lch * 0.5f * (g_y2 + g_y1), LHp = Lch * 0.5f * ((1 - g_y2) + (1 - g_y1));
        gP.w += LWg * aca + LHg * asa; gP.h += LWg * asa + LHg * aca; gP.a += LWg * dWg_a + LHg * dHg_a;
        gT.w += LWp * acb + LHp * asb; gT.h += LWp * asb + LHp * acb; gT.a += LWp * dWp_a + LHp * dHp_a;
    }

    // -- chain rule back to the spherical inputs (degrees) --
    planar_to_spherical_grads<DIM>(f, gP, gT, /*x_split=*/0.5f, pred, target, /*jitter=*/true, EDGE_ARC, gpred, gtarget);
    return loss;
}

// Adjoint of sph2pob_{standard,efficient}(...,'rad') (+ the two jitters when `jitter`): gradients of the two planar
// boxes (x, y, w, h, a) -> gradients of the spherical inputs.  y carries no gradient (it is the constant pi/2 | 0).
template <int VARIANT, int DIM>
SPH_DEV void pair_transform_bwd(const float (&in1)[5], const float (&in2)[5], const float (&g1)[5], const float (&g2)[5],
                                int edge, bool jitter, float (&gin1)[5], float (&gin2)[5]) {
    float b1[5], b2[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { b1[k] = in1[k]; b2[k] = in2[k]; }
    if (jitter) jitter_spherical<DIM>(b1, b2);
    LossFront f;
    loss_front_fast<DIM, VARIANT>(b1, b2, f, edge, jitter);
    PlanarGrad gP{g1[0], g1[2], g1[3], g1[4]}, gT{g2[0], g2[2], g2[3], g2[4]};
    planar_to_spherical_grads<DIM>(f, gP, gT, VARIANT == VARIANT_STANDARD ? 0.5f : 0.0f, in1, in2, jitter, edge, gin1, gin2);
}

}  // namespace sph2pob

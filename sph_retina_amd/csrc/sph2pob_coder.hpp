// Per-row arithmetic of the spherical delta box coders and of the OBB L1 loss body, shared by the gfx950 kernels
// (sph2pob_coder.hip) and by their CPU twins (sph2pob_host.hip): one source, so that a CPU tensor gets the operation order
// a device tensor gets.  Follows the reference's operation order (sphdet/bbox/coder/delta_xywh_sph_bbox_coder.py:137-161,
// :221-263; delta_xywha_rsph_bbox_coder.py:137-164, :224-268; sphdet/losses/sph2pob_l1_loss.py:28-88) so that results agree
// to the rounding of exp/log.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/sph2pob_hip.h"

namespace sph2pob_coder {

#define SPHC_DEV __host__ __device__ __forceinline__

constexpr int64_t kMaxElems = (int64_t)1 << 40;
constexpr float kEps = 1e-7f;                      // the coders' eps (:137, :221)
constexpr float kRad2Deg = 57.29577951308232f;     // torch.rad2deg multiplies by fp32(180/pi)
constexpr float kDeg2Rad = 0.017453292519943295f;  // torch.deg2rad multiplies by fp32(pi/180)
constexpr float kPiF = 3.14159265358979323846f;

struct Norm {
    float mean[5];
    float std[5];
};

inline Norm make_norm(const float* means, const float* stds, int dim) {
    Norm nm;
    for (int k = 0; k < 5; k++) {
        nm.mean[k] = (means && k < dim) ? means[k] : 0.0f;
        nm.std[k] = (stds && k < dim) ? stds[k] : 1.0f;
    }
    return nm;
}

// torch.clamp semantics: NaN propagates (fminf/fmaxf would drop it)
SPHC_DEV float clamp_lo(float x, float lo) { return x != x ? x : fmaxf(x, lo); }
SPHC_DEV float clamp_hi(float x, float hi) { return x != x ? x : fminf(x, hi); }
SPHC_DEV float clamp2(float x, float lo, float hi) { return x != x ? x : fminf(fmaxf(x, lo), hi); }
// gradient gate of torch.clamp: passes where lo <= x <= hi
SPHC_DEV float gate2(float x, float lo, float hi) { return (x >= lo && x <= hi) ? 1.0f : 0.0f; }

// upper clamp bounds as the reference forms them: python float (360 - 1e-7) etc. cast to fp32
SPHC_DEV float hi_theta() { return (float)(360.0 - 1e-7); }
SPHC_DEV float hi_half() { return (float)(180.0 - 1e-7); }
SPHC_DEV float lo_gamma() { return (float)(-90.0 + 1e-7); }
SPHC_DEV float hi_gamma() { return (float)(90.0 - 1e-7); }

// deltas of one proposal p w.r.t. its gt g (bbox2delta)
template <int DIM>
SPHC_DEV void encode_one(const float* p, const float* g, const Norm& nm, float* d) {
    float pw = clamp_lo(p[2], kEps), ph = clamp_lo(p[3], kEps);
    float gw = clamp_lo(g[2], kEps), gh = clamp_lo(g[3], kEps);
    d[0] = (g[0] - p[0]) / pw;
    d[1] = (g[1] - p[1]) / ph;
    d[2] = logf(gw / pw);
    d[3] = logf(gh / ph);
    if (DIM == 5) d[4] = (g[4] - p[4]) * kDeg2Rad;
#pragma unroll
    for (int k = 0; k < DIM; k++) d[k] = (d[k] - nm.mean[k]) / nm.std[k];
}

// One decoded box; when BWD, also the diagonal Jacobian d box[k] / d delta[k].
template <int DIM, bool BWD>
SPHC_DEV void decode_one(const float* p, const float* dl, const Norm& nm, float max_ratio, int flags,
                                           float ctr_clamp, float* box, float* jac) {
    float den[5];
#pragma unroll
    for (int k = 0; k < DIM; k++) den[k] = dl[k] * nm.std[k] + nm.mean[k];
    float sx = p[2] * den[0], sy = p[3] * den[1];
    float dw = den[2], dh = den[3];
    float gsx = 1.0f, gsy = 1.0f, gdw, gdh;
    if (flags & SPH2POB_CODER_CTR_CLAMP) {
        if (BWD) { gsx = gate2(sx, -ctr_clamp, ctr_clamp); gsy = gate2(sy, -ctr_clamp, ctr_clamp); }
        sx = clamp2(sx, -ctr_clamp, ctr_clamp);
        sy = clamp2(sy, -ctr_clamp, ctr_clamp);
        gdw = dw <= max_ratio ? 1.0f : 0.0f;
        gdh = dh <= max_ratio ? 1.0f : 0.0f;
        dw = clamp_hi(dw, max_ratio);
        dh = clamp_hi(dh, max_ratio);
    } else {
        gdw = gate2(dw, -max_ratio, max_ratio);
        gdh = gate2(dh, -max_ratio, max_ratio);
        dw = clamp2(dw, -max_ratio, max_ratio);
        dh = clamp2(dh, -max_ratio, max_ratio);
    }
    float x = p[0] + sx, y = p[1] + sy;
    float w = p[2] * expf(dw), h = p[3] * expf(dh);
    float a = 0.0f;
    if (DIM == 5) a = p[4] + den[4] * kRad2Deg;
    float bx = 1.0f, by = 1.0f, bw = 1.0f, bh = 1.0f, ba = 1.0f;
    if (flags & SPH2POB_CODER_CLIP_BORDER) {
        if (BWD) {
            bx = gate2(x, kEps, hi_theta()); by = gate2(y, kEps, hi_half());
            bw = gate2(w, kEps, hi_half()); bh = gate2(h, kEps, hi_half());
            if (DIM == 5) ba = gate2(a, lo_gamma(), hi_gamma());
        }
        x = clamp2(x, kEps, hi_theta());
        y = clamp2(y, kEps, hi_half());
        w = clamp2(w, kEps, hi_half());
        h = clamp2(h, kEps, hi_half());
        if (DIM == 5) a = clamp2(a, lo_gamma(), hi_gamma());
    }
    if (BWD) {
        // w, h before the border clamp are p * exp(d): reuse them through the gates (a clamped value has gate 0)
        jac[0] = bx * gsx * p[2] * nm.std[0];
        jac[1] = by * gsy * p[3] * nm.std[1];
        jac[2] = bw * gdw * (p[2] * expf(dw)) * nm.std[2];
        jac[3] = bh * gdh * (p[3] * expf(dh)) * nm.std[3];
        if (DIM == 5) jac[4] = ba * kRad2Deg * nm.std[4];
    }
    box[0] = x; box[1] = y; box[2] = w; box[3] = h;
    if (DIM == 5) box[4] = a;
}

// ---- OBB L1 loss body on planar boxes (sphdet/losses/sph2pob_l1_loss.py:28-88) ----
SPHC_DEV float wrap_angle(float a, bool modulus) {
    if (!modulus) return a;
    float r = fmodf(a + kPiF, kPiF);  // torch `%` is a floored remainder: result takes the sign of the divisor
    return (r != 0.0f && r < 0.0f) ? r + kPiF : r;
}
SPHC_DEV float sgn(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// deltas of proposals p w.r.t. gt g (bbox2delta :39-80, means 0 / stds 1)
SPHC_DEV void obb_deltas(const float* p, const float* g, bool modulus, float* d, float& pw, float& ph,
                                           float& gw, float& gh) {
    pw = clamp_lo(p[2], kEps); ph = clamp_lo(p[3], kEps);
    gw = clamp_lo(g[2], kEps); gh = clamp_lo(g[3], kEps);
    d[0] = (g[0] - p[0]) / pw;
    d[1] = (g[1] - p[1]) / ph;
    d[4] = (wrap_angle(g[4], modulus) - wrap_angle(p[4], modulus)) / kPiF;
    d[2] = logf(gw / pw);
    d[3] = logf(gh / ph);
}

// loss row of the L1 body: a = prediction, b = target (planar boxes), w = per-element weights
SPHC_DEV void l1_fwd_one(const float* a, const float* b, const float* w, float scale, int flags, float* d) {
    if (flags & SPH2POB_L1_ENCODE) {
        float pw, ph, gw, gh;
        const bool swap = flags & SPH2POB_L1_SWAP;
        obb_deltas(swap ? b : a, swap ? a : b, flags & SPH2POB_L1_MODULUS, d, pw, ph, gw, gh);
    } else {
#pragma unroll
        for (int k = 0; k < 5; k++) d[k] = a[k] - b[k];
    }
#pragma unroll
    for (int k = 0; k < 5; k++) d[k] = scale * (fabsf(d[k]) * w[k]);
}

// its adjoint: u = upstream gradient of the loss row (overwritten), ga / gb = gradients w.r.t. a / b
SPHC_DEV void l1_bwd_one(const float* a, const float* b, const float* w, float* u, float scale, int flags, float* ga, float* gb) {
    float d[5];
    if (flags & SPH2POB_L1_ENCODE) {
        const bool swap = flags & SPH2POB_L1_SWAP;
        const float* p = swap ? b : a;
        const float* g = swap ? a : b;
        float pw, ph, gw, gh, gp[5], gg[5];
        obb_deltas(p, g, flags & SPH2POB_L1_MODULUS, d, pw, ph, gw, gh);
#pragma unroll
        for (int k = 0; k < 5; k++) u[k] = (u[k] * scale) * w[k] * sgn(d[k]);
        gg[0] = u[0] / pw; gp[0] = -gg[0];
        gg[1] = u[1] / ph; gp[1] = -gg[1];
        // clip(min=eps) passes gradients where the width is >= eps
        gp[2] = p[2] >= kEps ? -(u[0] * (g[0] - p[0]) / pw) / pw - u[2] / pw : 0.0f;
        gp[3] = p[3] >= kEps ? -(u[1] * (g[1] - p[1]) / ph) / ph - u[3] / ph : 0.0f;
        gg[2] = g[2] >= kEps ? u[2] / gw : 0.0f;
        gg[3] = g[3] >= kEps ? u[3] / gh : 0.0f;
        gg[4] = u[4] / kPiF; gp[4] = -gg[4];
#pragma unroll
        for (int k = 0; k < 5; k++) { ga[k] = swap ? gg[k] : gp[k]; gb[k] = swap ? gp[k] : gg[k]; }
    } else {
#pragma unroll
        for (int k = 0; k < 5; k++) { ga[k] = (u[k] * scale) * w[k] * sgn(a[k] - b[k]); gb[k] = -ga[k]; }
    }
}

// argument checks shared by the HIP entry points and their CPU twins
inline int check_encode(const void* proposals, const void* gt, const void* deltas, int64_t n, int box_dim) {
    if (box_dim != 4 && box_dim != 5) return SPH2POB_ERR_DIM;
    if (n < 0 || n > kMaxElems) return SPH2POB_ERR_SIZE;
    if (n == 0) return 0;
    if (!proposals || !gt || !deltas) return SPH2POB_ERR_NULL;
    return 0;
}
inline int check_decode(const void* rois, const void* deltas, const void* grad, const void* out, int64_t n, int num_classes, int box_dim,
                        float max_ratio, int flags) {
    if (box_dim != 4 && box_dim != 5) return SPH2POB_ERR_DIM;
    if (num_classes < 1 || (flags & ~3) || !(max_ratio >= 0.0f)) return SPH2POB_ERR_OPTION;
    if (n < 0 || n > kMaxElems / num_classes) return SPH2POB_ERR_SIZE;
    if (n == 0) return 0;
    if (!rois || !deltas || !grad || !out) return SPH2POB_ERR_NULL;
    return 0;
}
inline int check_l1(const void* pred, const void* target, const void* grad, const void* out, int64_t n, int flags) {
    if (flags & ~7) return SPH2POB_ERR_OPTION;
    if (n < 0 || n > kMaxElems) return SPH2POB_ERR_SIZE;
    if (n == 0) return 0;
    if (!pred || !target || !grad || !out) return SPH2POB_ERR_NULL;
    return 0;
}

}  // namespace sph2pob_coder

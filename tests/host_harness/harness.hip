// TEST INFRASTRUCTURE (never shipped, never loaded by the product): compiles the product's device math
// (sph_retina_amd/csrc/*.hpp, __host__ __device__) for the HOST so that the hand-derived loss adjoint and the
// boundary-integral rectangle intersection can be unit-tested and sanitised on machines without a GPU.
#include <stdint.h>
#include "../../sph_retina_amd/csrc/sph2pob_device.hpp"
#include "../../sph_retina_amd/csrc/sph2pob_loss.hpp"
#include "../../sph_retina_amd/csrc/sph2pob_fast.hpp"
#include "../../sph_retina_amd/csrc/sph2pob_unbiased.hpp"

using namespace sph2pob;

template <int DIM, bool FAST>
static void loss_loop(const float* pred, const float* target, int64_t n, int mode, float eps, float* loss, float* iou,
                      float* gp, float* gt) {
    for (int64_t i = 0; i < n; i++) {
        float x[5] = {0, 0, 0, 0, 0}, y[5] = {0, 0, 0, 0, 0}, gx[5], gy[5], io;
        for (int k = 0; k < DIM; k++) { x[k] = pred[i * DIM + k]; y[k] = target[i * DIM + k]; }
        loss[i] = pair_loss<DIM, true, FAST>(x, y, mode, eps, &io, gx, gy);
        iou[i] = io;
        for (int k = 0; k < DIM; k++) { gp[i * DIM + k] = gx[k]; gt[i * DIM + k] = gy[k]; }
    }
}

template <int V, int DIM>
static void iou_loop(const float* b1, const float* b2, int64_t n, int mode, int edge, int angle, float* out) {
    for (int64_t i = 0; i < n; i++) {
        float x[5] = {0, 0, 0, 0, 0}, y[5] = {0, 0, 0, 0, 0};
        for (int k = 0; k < DIM; k++) { x[k] = b1[i * DIM + k]; y[k] = b2[i * DIM + k]; }
        out[i] = pair_iou<V, DIM>(x, y, mode, edge, angle);
    }
}

template <int V, int DIM>
static void iou_fast_loop(const float* b1, const float* b2, int64_t n, int mode, int edge, float* out) {
    for (int64_t i = 0; i < n; i++) {
        float x[5] = {0, 0, 0, 0, 0}, y[5] = {0, 0, 0, 0, 0};
        for (int k = 0; k < DIM; k++) { x[k] = b1[i * DIM + k]; y[k] = b2[i * DIM + k]; }
        out[i] = pair_iou_fast<V, DIM>(x, y, mode, edge);
    }
}

template <int V, int DIM>
static void tbwd_loop(const float* b1, const float* b2, const float* g1, const float* g2, int64_t n, int edge, int jitter,
                      float* o1, float* o2) {
    for (int64_t i = 0; i < n; i++) {
        float x[5] = {0, 0, 0, 0, 0}, y[5] = {0, 0, 0, 0, 0}, p[5], q[5], gx[5], gy[5];
        for (int k = 0; k < DIM; k++) { x[k] = b1[i * DIM + k]; y[k] = b2[i * DIM + k]; }
        for (int k = 0; k < 5; k++) { p[k] = g1[i * 5 + k]; q[k] = g2[i * 5 + k]; }
        pair_transform_bwd<V, DIM>(x, y, p, q, edge, jitter != 0, gx, gy);
        for (int k = 0; k < DIM; k++) { o1[i * DIM + k] = gx[k]; o2[i * DIM + k] = gy[k]; }
    }
}

template <int DIM>
static void extra_loop(const float* b1, const float* b2, int64_t n, int which, float* out) {
    for (int64_t i = 0; i < n; i++) {
        float x[5] = {0, 0, 0, 0, 0}, y[5] = {0, 0, 0, 0, 0};
        for (int k = 0; k < DIM; k++) { x[k] = b1[i * DIM + k]; y[k] = b2[i * DIM + k]; }
        out[i] = which == 0 ? unbiased_pair_iou<DIM, false>(x, y) : which == 1 ? unbiased_pair_iou<DIM, true>(x, y) : naive_iou<DIM>(x, y);
    }
}
extern "C" {
int harness_transform_bwd(const float* b1, const float* b2, const float* g1, const float* g2, int64_t n, int dim,
                          int variant, int edge, int jitter, float* o1, float* o2) {
    if (variant == 0) { if (dim == 4) tbwd_loop<0, 4>(b1, b2, g1, g2, n, edge, jitter, o1, o2); else tbwd_loop<0, 5>(b1, b2, g1, g2, n, edge, jitter, o1, o2); }
    else { if (dim == 4) tbwd_loop<1, 4>(b1, b2, g1, g2, n, edge, jitter, o1, o2); else tbwd_loop<1, 5>(b1, b2, g1, g2, n, edge, jitter, o1, o2); }
    return 0;
}
int harness_iou_fast(const float* b1, const float* b2, int64_t n, int dim, int variant, int mode, int edge, float* out) {
    if (variant == 0) { if (dim == 4) iou_fast_loop<0, 4>(b1, b2, n, mode, edge, out); else iou_fast_loop<0, 5>(b1, b2, n, mode, edge, out); }
    else { if (dim == 4) iou_fast_loop<1, 4>(b1, b2, n, mode, edge, out); else iou_fast_loop<1, 5>(b1, b2, n, mode, edge, out); }
    return 0;
}
int harness_loss(const float* pred, const float* target, int64_t n, int dim, int mode, float eps, float* loss,
                 float* iou, float* gp, float* gt, int fast) {
    if (fast) { if (dim == 4) loss_loop<4, true>(pred, target, n, mode, eps, loss, iou, gp, gt); else loss_loop<5, true>(pred, target, n, mode, eps, loss, iou, gp, gt); }
    else { if (dim == 4) loss_loop<4, false>(pred, target, n, mode, eps, loss, iou, gp, gt); else loss_loop<5, false>(pred, target, n, mode, eps, loss, iou, gp, gt); }
    return 0;
}
int harness_iou(const float* b1, const float* b2, int64_t n, int dim, int variant, int mode, int edge, int angle,
                float* out) {
    if (variant == 0) { if (dim == 4) iou_loop<0, 4>(b1, b2, n, mode, edge, angle, out); else iou_loop<0, 5>(b1, b2, n, mode, edge, angle, out); }
    else if (variant == 1) { if (dim == 4) iou_loop<1, 4>(b1, b2, n, mode, edge, angle, out); else iou_loop<1, 5>(b1, b2, n, mode, edge, angle, out); }
    else iou_loop<2, 4>(b1, b2, n, mode, edge, angle, out);
    return 0;
}
// which: 0 = unbiased IoU (double), 1 = unbiased IoU with the reference's fp32 roundings, 2 = naive IoU
int harness_extra_iou(const float* b1, const float* b2, int64_t n, int dim, int which, float* out) {
    if (dim == 4) extra_loop<4>(b1, b2, n, which, out); else extra_loop<5>(b1, b2, n, which, out);
    return 0;
}
// planar IoU through the near-parallel first-order routine of the closed-form path (test hook)
int harness_near_parallel_iou(const float* p1, const float* p2, int64_t n, float* out) {
    for (int64_t i = 0; i < n; i++) {
        const float* A = p1 + i * 5; const float* B = p2 + i * 5;
        float sa = sinf(A[4]), ca = cosf(A[4]), sb = sinf(B[4]), cb = cosf(B[4]);
        float c = ca * cb + sa * sb, s = sa * cb - ca * sb, dx = B[0] - A[0], dy = B[1] - A[1];
        float pax = -(dx * cb + dy * sb), pay = -(dy * cb - dx * sb);
        float inter = near_parallel_inter(pax, pay, c, s, 0.5f * A[2], 0.5f * A[3], 0.5f * B[2], 0.5f * B[3]);
        out[i] = inter / (A[2] * A[3] + B[2] * B[3] - inter);
    }
    return 0;
}
int harness_planar_iou(const float* p1, const float* p2, int64_t n, int mode, float* out) {
    for (int64_t i = 0; i < n; i++) {
        PBox A{p1[i * 5], p1[i * 5 + 1], p1[i * 5 + 2], p1[i * 5 + 3], p1[i * 5 + 4]};
        PBox B{p2[i * 5], p2[i * 5 + 1], p2[i * 5 + 2], p2[i * 5 + 3], p2[i * 5 + 4]};
        out[i] = planar_iou(A, B, mode);
    }
    return 0;
}
}


// adjoint of the reference-order transforms by forward-mode differentiation (legacy, rbb_angle='project')
template <int V, int DIM>
static void tbwd_dual_loop(const float* b1, const float* b2, const float* g1, const float* g2, int64_t n, int edge, int angle,
                           int jitter, float* o1, float* o2) {
    for (int64_t i = 0; i < n; i++) {
        float x[5] = {0, 0, 0, 0, 0}, y[5] = {0, 0, 0, 0, 0}, p[5], q[5], gx[5], gy[5];
        for (int k = 0; k < DIM; k++) { x[k] = b1[i * DIM + k]; y[k] = b2[i * DIM + k]; }
        for (int k = 0; k < 5; k++) { p[k] = g1[i * 5 + k]; q[k] = g2[i * 5 + k]; }
        transform_bwd_dual<V, DIM>(x, y, p, q, edge, angle, jitter != 0, gx, gy);
        for (int k = 0; k < DIM; k++) { o1[i * DIM + k] = gx[k]; o2[i * DIM + k] = gy[k]; }
    }
}
extern "C" int harness_transform_bwd_general(const float* b1, const float* b2, const float* g1, const float* g2, int64_t n,
                                             int dim, int variant, int edge, int angle, int jitter, float* o1, float* o2) {
    if (variant == 2) tbwd_dual_loop<2, 4>(b1, b2, g1, g2, n, edge, angle, jitter, o1, o2);
    else if (variant == 0) { if (dim == 4) tbwd_dual_loop<0, 4>(b1, b2, g1, g2, n, edge, angle, jitter, o1, o2); else tbwd_dual_loop<0, 5>(b1, b2, g1, g2, n, edge, angle, jitter, o1, o2); }
    else { if (dim == 4) tbwd_dual_loop<1, 4>(b1, b2, g1, g2, n, edge, angle, jitter, o1, o2); else tbwd_dual_loop<1, 5>(b1, b2, g1, g2, n, edge, angle, jitter, o1, o2); }
    return 0;
}

// stage-0 cull decisions (1 = culled): the arc form (standard / efficient) and the chord form (legacy)
extern "C" int harness_cull(const float* b1, const float* b2, int64_t n, int dim, int chord, unsigned char* out) {
    for (int64_t i = 0; i < n; i++) {
        float x[5] = {0, 0, 0, 0, 0}, y[5] = {0, 0, 0, 0, 0};
        for (int k = 0; k < dim; k++) { x[k] = b1[i * dim + k]; y[k] = b2[i * dim + k]; }
        bool c;
        if (dim == 4) c = chord ? fast_cull<4, true>(x, y, EDGE_ARC) : fast_cull<4, false>(x, y, EDGE_ARC);
        else c = chord ? fast_cull<5, true>(x, y, EDGE_ARC) : fast_cull<5, false>(x, y, EDGE_ARC);
        out[i] = c ? 1 : 0;
    }
    return 0;
}

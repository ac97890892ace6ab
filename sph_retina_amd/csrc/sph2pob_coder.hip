// Spherical delta box coder for gfx950: encode / decode / decode-adjoint, and the OBB L1 loss body.
//
// Element-wise and HBM-bound: one lane per (row, class) box, 16-byte accesses for BFoV, per-column accesses that a
// wave coalesces into whole cache lines for RBFoV (64 lanes x 20 B contiguous).  The per-row arithmetic lives in
// sph2pob_coder.hpp (shared with the CPU twins) and follows the reference's operation order
// (sphdet/bbox/coder/delta_xywh_sph_bbox_coder.py:137-161, :221-263 and delta_xywha_rsph_bbox_coder.py:137-164, :224-268)
// so that results agree to the rounding of exp/log.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sph2pob_hip.h"
#include "sph2pob_coder.hpp"

namespace {

using namespace sph2pob_coder;

constexpr int kBlock = 256;

template <int DIM>
__device__ __forceinline__ void load_row(const float* __restrict__ p, int64_t i, float* v) {
    if (DIM == 4) {
        float4 t = reinterpret_cast<const float4*>(p)[i];
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int k = 0; k < 5; k++) v[k] = p[i * 5 + k];
    }
}

template <int DIM>
__device__ __forceinline__ void store_row(float* __restrict__ p, int64_t i, const float* v) {
    if (DIM == 4) {
        reinterpret_cast<float4*>(p)[i] = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int k = 0; k < 5; k++) p[i * 5 + k] = v[k];
    }
}

template <int DIM>
__global__ __launch_bounds__(kBlock) void coder_encode_kernel(const float* __restrict__ proposals,
                                                             const float* __restrict__ gt, Norm nm,
                                                             float* __restrict__ deltas, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float p[5], g[5], d[5];
    load_row<DIM>(proposals, i, p);
    load_row<DIM>(gt, i, g);
    encode_one<DIM>(p, g, nm, d);
    store_row<DIM>(deltas, i, d);
}

template <int DIM>
__global__ __launch_bounds__(kBlock) void coder_decode_kernel(const float* __restrict__ rois,
                                                             const float* __restrict__ deltas, Norm nm,
                                                             float* __restrict__ boxes, int64_t total, int num_classes,
                                                             float max_ratio, int flags, float ctr_clamp) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= total) return;
    float p[5], d[5], b[5];
    load_row<DIM>(rois, num_classes == 1 ? i : i / num_classes, p);
    load_row<DIM>(deltas, i, d);
    decode_one<DIM, false>(p, d, nm, max_ratio, flags, ctr_clamp, b, nullptr);
    store_row<DIM>(boxes, i, b);
}

template <int DIM>
__global__ __launch_bounds__(kBlock) void coder_decode_bwd_kernel(const float* __restrict__ rois,
                                                                 const float* __restrict__ deltas,
                                                                 const float* __restrict__ grad_boxes, Norm nm,
                                                                 float* __restrict__ grad_deltas, int64_t total,
                                                                 int num_classes, float max_ratio, int flags,
                                                                 float ctr_clamp) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= total) return;
    float p[5], d[5], b[5], j[5], g[5];
    load_row<DIM>(rois, num_classes == 1 ? i : i / num_classes, p);
    load_row<DIM>(deltas, i, d);
    load_row<DIM>(grad_boxes, i, g);
    decode_one<DIM, true>(p, d, nm, max_ratio, flags, ctr_clamp, b, j);
#pragma unroll
    for (int k = 0; k < DIM; k++) g[k] = g[k] * j[k];
    store_row<DIM>(grad_deltas, i, g);
}

// ---- OBB L1 loss body on planar boxes (sphdet/losses/sph2pob_l1_loss.py:28-88) ----
__global__ __launch_bounds__(kBlock) void obb_l1_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                           const float* __restrict__ weight, float scale,
                                                           float* __restrict__ loss, int64_t n, int flags) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float a[5], b[5], d[5], w[5] = {1.0f, 1.0f, 1.0f, 1.0f, 1.0f};
    load_row<5>(pred, i, a);
    load_row<5>(target, i, b);
    if (weight) load_row<5>(weight, i, w);
    l1_fwd_one(a, b, w, scale, flags, d);
    store_row<5>(loss, i, d);
}

__global__ __launch_bounds__(kBlock) void obb_l1_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                           const float* __restrict__ weight,
                                                           const float* __restrict__ grad_loss, float scale,
                                                           float* __restrict__ grad_pred, float* __restrict__ grad_target,
                                                           int64_t n, int flags) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float a[5], b[5], u[5], w[5] = {1.0f, 1.0f, 1.0f, 1.0f, 1.0f}, ga[5], gb[5];
    load_row<5>(pred, i, a);
    load_row<5>(target, i, b);
    load_row<5>(grad_loss, i, u);
    if (weight) load_row<5>(weight, i, w);
    l1_bwd_one(a, b, w, u, scale, flags, ga, gb);
    store_row<5>(grad_pred, i, ga);
    if (grad_target) store_row<5>(grad_target, i, gb);
}

int status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

}  // namespace

extern "C" {

int sph2pob_coder_encode_f32(const float* proposals, const float* gt, const float* means_host, const float* stds_host,
                             float* deltas, int64_t n, int box_dim, void* stream) {
    if (int rc = check_encode(proposals, gt, deltas, n, box_dim)) return rc;
    if (n == 0) return 0;
    Norm nm = make_norm(means_host, stds_host, box_dim);
    dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    if (box_dim == 4)
        hipLaunchKernelGGL((coder_encode_kernel<4>), grid, dim3(kBlock), 0, (hipStream_t)stream, proposals, gt, nm, deltas, n);
    else
        hipLaunchKernelGGL((coder_encode_kernel<5>), grid, dim3(kBlock), 0, (hipStream_t)stream, proposals, gt, nm, deltas, n);
    return status();
}

int sph2pob_coder_decode_f32(const float* rois, const float* deltas, const float* means_host, const float* stds_host,
                             float* boxes, int64_t n, int num_classes, int box_dim, float max_ratio, int flags,
                             float ctr_clamp, void* stream) {
    if (int rc = check_decode(rois, deltas, deltas, boxes, n, num_classes, box_dim, max_ratio, flags)) return rc;
    if (n == 0) return 0;
    Norm nm = make_norm(means_host, stds_host, box_dim);
    int64_t total = n * num_classes;
    dim3 grid((unsigned)((total + kBlock - 1) / kBlock));
    if (box_dim == 4)
        hipLaunchKernelGGL((coder_decode_kernel<4>), grid, dim3(kBlock), 0, (hipStream_t)stream, rois, deltas, nm, boxes, total, num_classes, max_ratio, flags, ctr_clamp);
    else
        hipLaunchKernelGGL((coder_decode_kernel<5>), grid, dim3(kBlock), 0, (hipStream_t)stream, rois, deltas, nm, boxes, total, num_classes, max_ratio, flags, ctr_clamp);
    return status();
}

int sph2pob_coder_decode_bwd_f32(const float* rois, const float* deltas, const float* grad_boxes,
                                 const float* means_host, const float* stds_host, float* grad_deltas, int64_t n,
                                 int num_classes, int box_dim, float max_ratio, int flags, float ctr_clamp,
                                 void* stream) {
    if (int rc = check_decode(rois, deltas, grad_boxes, grad_deltas, n, num_classes, box_dim, max_ratio, flags)) return rc;
    if (n == 0) return 0;
    Norm nm = make_norm(means_host, stds_host, box_dim);
    int64_t total = n * num_classes;
    dim3 grid((unsigned)((total + kBlock - 1) / kBlock));
    if (box_dim == 4)
        hipLaunchKernelGGL((coder_decode_bwd_kernel<4>), grid, dim3(kBlock), 0, (hipStream_t)stream, rois, deltas, grad_boxes, nm, grad_deltas, total, num_classes, max_ratio, flags, ctr_clamp);
    else
        hipLaunchKernelGGL((coder_decode_bwd_kernel<5>), grid, dim3(kBlock), 0, (hipStream_t)stream, rois, deltas, grad_boxes, nm, grad_deltas, total, num_classes, max_ratio, flags, ctr_clamp);
    return status();
}

int sph2pob_obb_l1_fwd_f32(const float* planar_pred, const float* planar_target, const float* weight, float scale,
                           float* loss, int64_t n, int flags, void* stream) {
    if (int rc = check_l1(planar_pred, planar_target, planar_pred, loss, n, flags)) return rc;
    if (n == 0) return 0;
    dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(obb_l1_fwd_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, planar_pred, planar_target, weight, scale, loss, n, flags);
    return status();
}

int sph2pob_obb_l1_bwd_f32(const float* planar_pred, const float* planar_target, const float* weight,
                           const float* grad_loss, float scale, float* grad_pred, float* grad_target, int64_t n,
                           int flags, void* stream) {
    if (int rc = check_l1(planar_pred, planar_target, grad_loss, grad_pred, n, flags)) return rc;
    if (n == 0) return 0;
    dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(obb_l1_bwd_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, planar_pred, planar_target, weight, grad_loss, scale, grad_pred, grad_target, n, flags);
    return status();
}

}  // extern "C"

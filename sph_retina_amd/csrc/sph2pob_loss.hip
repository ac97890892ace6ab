// libsph2pob_hip.so — IoU / GIoU / DIoU / CIoU loss forward + backward and the deterministic sum: kernels + C-ABI launchers
// (include/sph2pob_hip.h).  gfx950 only.

#include "sph2pob_kernels_common.hpp"

namespace {


// ---- loss: per-element weight = mean over weight_dim columns (reference: sph2pob_transform.py:32-34 widens a
// (n,4) weight with its own mean, OBBIoULoss.forward then takes weight.mean(-1): sph2pob_iou_loss.py:48) ----
template <int DIM>
__device__ __forceinline__ float element_weight(const float* __restrict__ w, int wd, int64_t i) {
    if (!w) return 1.0f;
    if (wd == 1) return w[i];
    float v[DIM], s = 0.0f;   // wd == DIM here (the launchers reject anything else): DIM loads in flight, not a loop of load + wait
#pragma unroll
    for (int k = 0; k < DIM; k++) v[k] = w[i * DIM + k];
#pragma unroll
    for (int k = 0; k < DIM; k++) s += v[k];
    if (DIM == 4) return (s + s / 4.0f) / 5.0f;
    return s / (float)DIM;
}

// waves per SIMD the loss kernels that carry the adjoint are compiled for (closed-form front end): 4 = ~105 VGPRs, no
// scratch; 5 = 96 VGPRs with 2-5 spilled dwords; 6 = 80 VGPRs with ~20.  Measured on MI355X, 1 M RBFoV pairs, CIoU
// forward + backward through the C ABI: 30.5-30.8 us / 31.2-31.9 us / 41.6 us (profiles/r02t_loss_waves.log): 4 stays.
#if !defined(SPH_LOSS_WAVES)
#define SPH_LOSS_WAVES 4
#endif
constexpr int kLossWaves = SPH_LOSS_WAVES;
template <int DIM, bool FAST>
__global__ __launch_bounds__(kBlock) void loss_fwd_kernel(const float* __restrict__ pred,
                                                         const float* __restrict__ target,
                                                         const float* __restrict__ weight, int wd,
                                                         float scale, float* __restrict__ loss,
                                                         float* __restrict__ iou, int64_t n, int loss_mode,
                                                         float eps) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float x[5], y[5], gx[5], gy[5], io;
    const float w = scale * element_weight<DIM>(weight, wd, i);
    // dense heads pass every anchor with weight 0 on the negatives (sph_retina_head.py:261-264): a wave whose 64
    // weights are all zero writes its zeros and leaves (loss * 0 == 0 for every finite loss)
    if (!iou && __builtin_amdgcn_ballot_w64(w != 0.0f) == 0) {
        loss[i] = 0.0f;
        return;
    }
    load_box<DIM>(pred, i, x);
    load_box<DIM>(target, i, y);
    float l = pair_loss<DIM, false, FAST>(x, y, loss_mode, eps, &io, gx, gy);
    loss[i] = l * w;
    if (iou) iou[i] = io;
}

__device__ __forceinline__ float block_sum(float v);
// forward + per-workgroup partial sum (reduction 'mean' / 'sum'): no element buffer
template <int DIM, bool FAST>
__global__ __launch_bounds__(kBlock) void loss_fwd_sum_kernel(const float* __restrict__ pred,
                                                             const float* __restrict__ target,
                                                             const float* __restrict__ weight, int wd,
                                                             float* __restrict__ partial, int64_t n, int loss_mode,
                                                             float eps) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    float v = 0.0f;
    const bool live = i < n;
    const float w = live ? element_weight<DIM>(weight, wd, i) : 0.0f;
    if (__builtin_amdgcn_ballot_w64(w != 0.0f) != 0) {   // an all-zero-weight wave contributes exact zeros (see loss_fwd_kernel)
        if (live) {
            float x[5], y[5], gx[5], gy[5];
            load_box<DIM>(pred, i, x);
            load_box<DIM>(target, i, y);
            v = pair_loss<DIM, false, FAST>(x, y, loss_mode, eps, nullptr, gx, gy) * w;
        }
    }
    const float r = block_sum(v);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// GT: the caller wants the target's gradient too.  A training step does not (the target carries no gradient): with GT =
// false the target half of the chain rule back to the spherical inputs is dead code and the compiler drops it.
template <int DIM, bool FAST, bool GT>
__global__ __launch_bounds__(kBlock, FAST ? kLossWaves : 4) void loss_bwd_kernel(const float* __restrict__ pred,
                                                         const float* __restrict__ target,
                                                         const float* __restrict__ weight, int wd,
                                                         const float* __restrict__ grad_out, int grad_stride,
                                                         float scale, float* __restrict__ gpred,
                                                         float* __restrict__ gtarget, int64_t n, int loss_mode,
                                                         float eps) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float x[5], y[5], gx[5], gy[5];
    float g = grad_out[i * grad_stride] * scale * element_weight<DIM>(weight, wd, i);
    if (__builtin_amdgcn_ballot_w64(g != 0.0f) == 0) {  // all-negative wave (see loss_fwd_kernel): zero gradients, no geometry
#pragma unroll
        for (int k = 0; k < 5; k++) gx[k] = gy[k] = 0.0f;
    } else {
        load_box<DIM>(pred, i, x);
        load_box<DIM>(target, i, y);
        pair_loss<DIM, true, FAST>(x, y, loss_mode, eps, nullptr, gx, gy);
    }
    if (DIM == 4) {
        reinterpret_cast<float4*>(gpred)[i] = make_float4(g * gx[0], g * gx[1], g * gx[2], g * gx[3]);
        if (GT) reinterpret_cast<float4*>(gtarget)[i] = make_float4(g * gy[0], g * gy[1], g * gy[2], g * gy[3]);
    } else {
#pragma unroll
        for (int k = 0; k < 5; k++) gpred[i * 5 + k] = g * gx[k];
        if (GT) {
#pragma unroll
            for (int k = 0; k < 5; k++) gtarget[i * 5 + k] = g * gy[k];
        }
    }
}

// forward + gradients for an upstream gradient of 1 (+ per-workgroup partial sum of the loss when `partial`)
template <int DIM, bool FAST, bool GT>
__global__ __launch_bounds__(kBlock, FAST ? kLossWaves : 4) void loss_fwd_grad_kernel(const float* __restrict__ pred,
                                                              const float* __restrict__ target,
                                                              const float* __restrict__ weight, int wd, float scale,
                                                              float* __restrict__ loss, float* __restrict__ partial,
                                                              float* __restrict__ gpred, float* __restrict__ gtarget,
                                                              int64_t n, int loss_mode, float eps) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < n;
    const float w = live ? scale * element_weight<DIM>(weight, wd, i) : 0.0f;
    float x[5], y[5], gx[5], gy[5], l = 0.0f;
#pragma unroll
    for (int k = 0; k < 5; k++) gx[k] = gy[k] = 0.0f;
    if (__builtin_amdgcn_ballot_w64(w != 0.0f) != 0) {   // an all-zero-weight wave: zero loss, zero gradients, no geometry
        if (live) {
            load_box<DIM>(pred, i, x);
            load_box<DIM>(target, i, y);
            l = pair_loss<DIM, true, FAST>(x, y, loss_mode, eps, nullptr, gx, gy) * w;
        }
    }
    if (live) {
        if (loss) loss[i] = l;
        if (DIM == 4) {
            reinterpret_cast<float4*>(gpred)[i] = make_float4(w * gx[0], w * gx[1], w * gx[2], w * gx[3]);
            if (GT) reinterpret_cast<float4*>(gtarget)[i] = make_float4(w * gy[0], w * gy[1], w * gy[2], w * gy[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 5; k++) gpred[i * 5 + k] = w * gx[k];
            if (GT) {
#pragma unroll
                for (int k = 0; k < 5; k++) gtarget[i * 5 + k] = w * gy[k];
            }
        }
    }
    if (partial) {   // workgroup-uniform
        const float r = block_sum(live ? l : 0.0f);
        if (threadIdx.x == 0) partial[blockIdx.x] = r;
    }
}
// out[i, :] = stash[i, :] * g[i * stride]: the whole of torch's backward after loss_fwd_grad_kernel
__global__ __launch_bounds__(kBlock) void grad_scale_kernel(const float* stash, const float* __restrict__ g, int stride, float* out,
                                                           int64_t total, int dim) {   // out may alias stash (in place)
    // in place with an upstream gradient of exactly 1 (a plain `loss.backward()`): the stash already is the gradient —
    // one scalar load per workgroup instead of a 40 MB pass (the stash tensor itself is handed to autograd).  The grid is
    // capped and strided so that this early exit costs a small launch, not the dispatch of 20 000 workgroups.
    if (stride == 0 && out == stash && g[0] == 1.0f) return;
    // (four elements in flight per lane instead of one: measured, no gain — the 40 MB pass is bandwidth-bound; r03p_ab_loss.log)
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock)
        out[e] = stash[e] * g[stride ? (e / dim) : 0];
}

// ---- deterministic two-pass sum (bitwise reproducible losses; no float atomics) ----
constexpr int kSumBlocks = 1024;
__device__ __forceinline__ float block_sum(float v) {
    __shared__ float sm[kBlock / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.0f;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < kBlock / 64; k++) r += sm[k];
    }
    return r;
}
__global__ __launch_bounds__(kBlock) void sum_pass1(const float* __restrict__ x, int64_t n, float* __restrict__ ws) {
    float acc = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) acc += x[i];
    float r = block_sum(acc);
    if (threadIdx.x == 0) ws[blockIdx.x] = r;
}
__global__ __launch_bounds__(kBlock) void sum_pass2(const float* __restrict__ ws, int nb, float scale,
                                                   float* __restrict__ out) {
    // 16 loads per round on clamped indices, all in flight before the first add (a loop of load + add waits for every
    // element: 1 M pairs leave 3 907 partials = 16 serial round trips per thread); the adds keep their order, a masked
    // element adds +0
    float acc = 0.0f;
    for (int i0 = threadIdx.x; i0 < nb; i0 += kBlock * 16) {
        float v[16];
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int i = i0 + t * kBlock;
            v[t] = ws[i < nb ? i : nb - 1];
        }
#pragma unroll
        for (int t = 0; t < 16; t++) acc += (i0 + t * kBlock < nb) ? v[t] : 0.0f;
    }
    float r = block_sum(acc);
    if (threadIdx.x == 0) out[0] = r * scale;
}


}  // namespace

extern "C" {


int sph2pob_loss_fwd_f32(const float* pred, const float* target, const float* weight, int weight_dim, float scale,
                         float* loss, float* iou, int64_t n, int box_dim, int loss_mode_flags, float eps, void* stream) {
    const int loss_mode = loss_mode_flags & 0xff;
    const bool fast = !(loss_mode_flags & SPH2POB_FLAG_REFERENCE_ORDER);
    if (loss_mode_flags & ~(0xff | SPH2POB_FLAG_REFERENCE_ORDER)) return SPH2POB_ERR_OPTION;
    if (box_dim != 4 && box_dim != 5) return SPH2POB_ERR_DIM;
    if (loss_mode < 0 || loss_mode > 3) return SPH2POB_ERR_OPTION;
    if (weight && weight_dim != 1 && weight_dim != box_dim) return SPH2POB_ERR_OPTION;
    if (n < 0 || n > kMaxElems) return SPH2POB_ERR_SIZE;
    if (n == 0) return SPH2POB_OK;
    if (!pred || !target || !loss) return SPH2POB_ERR_NULL;
    dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    hipStream_t s = (hipStream_t)stream;
#define SPH_LOSS_FWD(D, F) \
    hipLaunchKernelGGL((loss_fwd_kernel<D, F>), grid, dim3(kBlock), 0, s, pred, target, weight, weight_dim, scale, loss, iou, n, loss_mode, eps)
    if (box_dim == 4) { if (fast) SPH_LOSS_FWD(4, true); else SPH_LOSS_FWD(4, false); }
    else { if (fast) SPH_LOSS_FWD(5, true); else SPH_LOSS_FWD(5, false); }
#undef SPH_LOSS_FWD
    return launch_status();
}

int sph2pob_loss_bwd_f32(const float* pred, const float* target, const float* weight, int weight_dim,
                         const float* grad_out, int grad_stride, float scale, float* grad_pred, float* grad_target,
                         int64_t n, int box_dim, int loss_mode_flags, float eps, void* stream) {
    const int loss_mode = loss_mode_flags & 0xff;
    const bool fast = !(loss_mode_flags & SPH2POB_FLAG_REFERENCE_ORDER);
    if (loss_mode_flags & ~(0xff | SPH2POB_FLAG_REFERENCE_ORDER)) return SPH2POB_ERR_OPTION;
    if (box_dim != 4 && box_dim != 5) return SPH2POB_ERR_DIM;
    if (loss_mode < 0 || loss_mode > 3 || (grad_stride != 0 && grad_stride != 1)) return SPH2POB_ERR_OPTION;
    if (weight && weight_dim != 1 && weight_dim != box_dim) return SPH2POB_ERR_OPTION;
    if (n < 0 || n > kMaxElems) return SPH2POB_ERR_SIZE;
    if (n == 0) return SPH2POB_OK;
    if (!pred || !target || !grad_out || !grad_pred) return SPH2POB_ERR_NULL;
    dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    hipStream_t s = (hipStream_t)stream;
#define SPH_LOSS_BWD(D, F) \
    do { if (grad_target) hipLaunchKernelGGL((loss_bwd_kernel<D, F, true>), grid, dim3(kBlock), 0, s, pred, target, weight, weight_dim, grad_out, grad_stride, scale, grad_pred, grad_target, n, loss_mode, eps); \
         else hipLaunchKernelGGL((loss_bwd_kernel<D, F, false>), grid, dim3(kBlock), 0, s, pred, target, weight, weight_dim, grad_out, grad_stride, scale, grad_pred, grad_target, n, loss_mode, eps); } while (0)
    if (box_dim == 4) { if (fast) SPH_LOSS_BWD(4, true); else SPH_LOSS_BWD(4, false); }
    else { if (fast) SPH_LOSS_BWD(5, true); else SPH_LOSS_BWD(5, false); }
#undef SPH_LOSS_BWD
    return launch_status();
}

int64_t sph2pob_loss_sum_workspace_floats(int64_t n) { return (n + kBlock - 1) / kBlock + kSumBlocks; }

int sph2pob_loss_fwd_sum_f32(const float* pred, const float* target, const float* weight, int weight_dim, float scale,
                             float* out, float* workspace, int64_t n, int box_dim, int loss_mode_flags, float eps,
                             void* stream) {
    const int loss_mode = loss_mode_flags & 0xff;
    const bool fast = !(loss_mode_flags & SPH2POB_FLAG_REFERENCE_ORDER);
    if (loss_mode_flags & ~(0xff | SPH2POB_FLAG_REFERENCE_ORDER)) return SPH2POB_ERR_OPTION;
    if (box_dim != 4 && box_dim != 5) return SPH2POB_ERR_DIM;
    if (loss_mode < 0 || loss_mode > 3) return SPH2POB_ERR_OPTION;
    if (weight && weight_dim != 1 && weight_dim != box_dim) return SPH2POB_ERR_OPTION;
    if (n < 0 || n > kMaxElems) return SPH2POB_ERR_SIZE;
    if (!out || !workspace || (n > 0 && (!pred || !target))) return SPH2POB_ERR_NULL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t nb = (n + kBlock - 1) / kBlock;
    if (nb > 0) {
        dim3 grid((unsigned)nb);
#define SPH_LOSS_FWDS(D, F) \
        hipLaunchKernelGGL((loss_fwd_sum_kernel<D, F>), grid, dim3(kBlock), 0, s, pred, target, weight, weight_dim, workspace, n, loss_mode, eps)
        if (box_dim == 4) { if (fast) SPH_LOSS_FWDS(4, true); else SPH_LOSS_FWDS(4, false); }
        else { if (fast) SPH_LOSS_FWDS(5, true); else SPH_LOSS_FWDS(5, false); }
#undef SPH_LOSS_FWDS
    }
    if (nb <= 65536) {   // one workgroup adds the partials in a fixed order
        hipLaunchKernelGGL(sum_pass2, dim3(1), dim3(kBlock), 0, s, workspace, (int)nb, scale, out);
    } else {             // very large batches: the two-pass tree over the partials
        float* ws2 = workspace + nb;
        hipLaunchKernelGGL(sum_pass1, dim3(kSumBlocks), dim3(kBlock), 0, s, workspace, nb, ws2);
        hipLaunchKernelGGL(sum_pass2, dim3(1), dim3(kBlock), 0, s, ws2, kSumBlocks, scale, out);
    }
    return launch_status();
}

int sph2pob_loss_fwd_grad_f32(const float* pred, const float* target, const float* weight, int weight_dim, float scale,
                              float* loss, float* out_sum, float* workspace, float* grad_pred, float* grad_target,
                              int64_t n, int box_dim, int loss_mode_flags, float eps, void* stream) {
    const int loss_mode = loss_mode_flags & 0xff;
    const bool fast = !(loss_mode_flags & SPH2POB_FLAG_REFERENCE_ORDER);
    if (loss_mode_flags & ~(0xff | SPH2POB_FLAG_REFERENCE_ORDER)) return SPH2POB_ERR_OPTION;
    if (box_dim != 4 && box_dim != 5) return SPH2POB_ERR_DIM;
    if (loss_mode < 0 || loss_mode > 3) return SPH2POB_ERR_OPTION;
    if (weight && weight_dim != 1 && weight_dim != box_dim) return SPH2POB_ERR_OPTION;
    if (n < 0 || n > kMaxElems) return SPH2POB_ERR_SIZE;
    if ((out_sum && !workspace) || (n > 0 && (!pred || !target || !grad_pred))) return SPH2POB_ERR_NULL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t nb = (n + kBlock - 1) / kBlock;
    float* partial = out_sum ? workspace : nullptr;
    if (nb > 0) {
        dim3 grid((unsigned)nb);
#define SPH_LOSS_FG(D, F) \
        do { if (grad_target) hipLaunchKernelGGL((loss_fwd_grad_kernel<D, F, true>), grid, dim3(kBlock), 0, s, pred, target, weight, weight_dim, scale, loss, partial, grad_pred, grad_target, n, loss_mode, eps); \
             else hipLaunchKernelGGL((loss_fwd_grad_kernel<D, F, false>), grid, dim3(kBlock), 0, s, pred, target, weight, weight_dim, scale, loss, partial, grad_pred, grad_target, n, loss_mode, eps); } while (0)
        if (box_dim == 4) { if (fast) SPH_LOSS_FG(4, true); else SPH_LOSS_FG(4, false); }
        else { if (fast) SPH_LOSS_FG(5, true); else SPH_LOSS_FG(5, false); }
#undef SPH_LOSS_FG
    }
    if (out_sum) {   // scale is already inside the elements
        if (nb <= 65536) {
            hipLaunchKernelGGL(sum_pass2, dim3(1), dim3(kBlock), 0, s, workspace, (int)nb, 1.0f, out_sum);
        } else {
            float* ws2 = workspace + nb;
            hipLaunchKernelGGL(sum_pass1, dim3(kSumBlocks), dim3(kBlock), 0, s, workspace, nb, ws2);
            hipLaunchKernelGGL(sum_pass2, dim3(1), dim3(kBlock), 0, s, ws2, kSumBlocks, 1.0f, out_sum);
        }
    }
    return launch_status();
}

int sph2pob_loss_grad_scale_f32(const float* stash, const float* grad_out, int grad_stride, float* out, int64_t n,
                                int box_dim, void* stream) {
    if (box_dim != 4 && box_dim != 5) return SPH2POB_ERR_DIM;
    if (grad_stride != 0 && grad_stride != 1) return SPH2POB_ERR_OPTION;
    if (n < 0 || n > kMaxElems) return SPH2POB_ERR_SIZE;
    if (n == 0) return SPH2POB_OK;
    if (!stash || !grad_out || !out) return SPH2POB_ERR_NULL;
    const int64_t total = n * box_dim;
    int64_t blocks = (total + kBlock - 1) / kBlock;
    const int64_t cap = (int64_t)cu_count() * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(grad_scale_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream,
                       stash, grad_out, grad_stride, out, total, box_dim);
    return launch_status();
}

int sph2pob_sum_workspace_floats(void) { return kSumBlocks; }

int sph2pob_sum_f32(const float* x, int64_t n, float scale, float* out, float* workspace, void* stream) {
    if (n < 0 || n > kMaxElems) return SPH2POB_ERR_SIZE;
    if (!out || !workspace || (n > 0 && !x)) return SPH2POB_ERR_NULL;
    hipStream_t s = (hipStream_t)stream;
    int nb = (int)((n + kBlock - 1) / kBlock);
    if (nb > kSumBlocks) nb = kSumBlocks;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(sum_pass1, dim3(nb), dim3(kBlock), 0, s, x, n, workspace);
    hipLaunchKernelGGL(sum_pass2, dim3(1), dim3(kBlock), 0, s, workspace, nb, scale, out);
    return launch_status();
}

}  // extern "C"

// libsph2pob_hip.so — kernels + C-ABI launchers (see include/sph2pob_hip.h).  gfx950 only.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sph2pob_hip.h"
#include "sph2pob_device.hpp"

namespace {

using namespace sph2pob;

constexpr int kBlock = 256;  // 4 waves of 64 lanes

template <int DIM>
__device__ __forceinline__ void load_box(const float* __restrict__ p, int64_t i, float (&b)[5]) {
    if (DIM == 4) {  // one 16-byte load per lane: 1 KiB per wave instruction, fully coalesced
        float4 v = reinterpret_cast<const float4*>(p)[i];
        b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w; b[4] = 0.0f;
    } else {
        const float* q = p + i * 5;
#pragma unroll
        for (int k = 0; k < 5; k++) b[k] = q[k];
    }
}

template <int VARIANT, int DIM>
__global__ __launch_bounds__(kBlock) void iou_aligned_kernel(const float* __restrict__ b1,
                                                            const float* __restrict__ b2,
                                                            float* __restrict__ out, int64_t n, int mode, int edge,
                                                            int angle) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float x[5], y[5];
    load_box<DIM>(b1, i, x);
    load_box<DIM>(b2, i, y);
    out[i] = pair_iou<VARIANT, DIM>(x, y, mode, edge, angle);
}

// out[i*n + j]: consecutive lanes walk j (coalesced stores, b2 loads coalesced, b1 row is a broadcast).
template <int VARIANT, int DIM>
__global__ __launch_bounds__(kBlock) void iou_pairwise_kernel(const float* __restrict__ b1, int64_t m,
                                                             const float* __restrict__ b2, int64_t n,
                                                             float* __restrict__ out, int mode, int edge,
                                                             int angle) {
    int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    int64_t i = blockIdx.y;
    if (j >= n) return;
    float x[5], y[5];
    load_box<DIM>(b1, i, x);
    load_box<DIM>(b2, j, y);
    out[i * n + j] = pair_iou<VARIANT, DIM>(x, y, mode, edge, angle);
}

template <int VARIANT, int DIM>
__global__ __launch_bounds__(kBlock) void transform_kernel(const float* __restrict__ b1,
                                                          const float* __restrict__ b2, float* __restrict__ o1,
                                                          float* __restrict__ o2, int64_t n, int edge, int angle,
                                                          int jitter) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float x[5], y[5];
    load_box<DIM>(b1, i, x);
    load_box<DIM>(b2, i, y);
    if (jitter) jitter_spherical<DIM>(x, y);
    PBox p1, p2;
    transform<VARIANT, DIM>(x, y, edge, angle, p1, p2);
    if (jitter) jitter_rotated(p1, p2);
    float* q1 = o1 + i * 5;
    float* q2 = o2 + i * 5;
    q1[0] = p1.x; q1[1] = p1.y; q1[2] = p1.w; q1[3] = p1.h; q1[4] = p1.a;
    q2[0] = p2.x; q2[1] = p2.y; q2[2] = p2.w; q2[3] = p2.h; q2[4] = p2.a;
}

int check_common(int box_dim, int variant, int edge, int angle) {
    if (box_dim != 4 && box_dim != 5) return SPH2POB_ERR_DIM;
    if (variant < 0 || variant > 2 || edge < 0 || edge > 2 || angle < 0 || angle > 1) return SPH2POB_ERR_OPTION;
    if (variant == SPH2POB_VARIANT_LEGACY && box_dim == 5) return SPH2POB_ERR_DIM;
    return SPH2POB_OK;
}

int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SPH2POB_OK : (int)e;
}

// dispatch a (VARIANT, DIM) pair to a functor
template <typename F>
int dispatch(int variant, int box_dim, F&& f) {
    if (variant == SPH2POB_VARIANT_STANDARD) return box_dim == 4 ? f.template run<0, 4>() : f.template run<0, 5>();
    if (variant == SPH2POB_VARIANT_EFFICIENT) return box_dim == 4 ? f.template run<1, 4>() : f.template run<1, 5>();
    return f.template run<2, 4>();
}

struct AlignedLaunch {
    const float *b1, *b2; float* out; int64_t n; int mode, edge, angle; hipStream_t s;
    template <int V, int D> int run() {
        dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
        hipLaunchKernelGGL((iou_aligned_kernel<V, D>), grid, dim3(kBlock), 0, s, b1, b2, out, n, mode, edge, angle);
        return launch_status();
    }
};
struct PairwiseLaunch {
    const float* b1; int64_t m; const float* b2; int64_t n; float* out; int mode, edge, angle; hipStream_t s;
    template <int V, int D> int run() {
        // grid.y is limited to 65535 rows per launch: walk the rows in slabs
        const int64_t kMaxRows = 65535;
        for (int64_t r0 = 0; r0 < m; r0 += kMaxRows) {
            int64_t rows = m - r0 < kMaxRows ? m - r0 : kMaxRows;
            dim3 grid((unsigned)((n + kBlock - 1) / kBlock), (unsigned)rows);
            hipLaunchKernelGGL((iou_pairwise_kernel<V, D>), grid, dim3(kBlock), 0, s, b1 + r0 * D, rows, b2, n,
                               out + r0 * n, mode, edge, angle);
            int rc = launch_status();
            if (rc) return rc;
        }
        return SPH2POB_OK;
    }
};
struct TransformLaunch {
    const float *b1, *b2; float *o1, *o2; int64_t n; int edge, angle, jitter; hipStream_t s;
    template <int V, int D> int run() {
        dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
        hipLaunchKernelGGL((transform_kernel<V, D>), grid, dim3(kBlock), 0, s, b1, b2, o1, o2, n, edge, angle, jitter);
        return launch_status();
    }
};

constexpr int64_t kMaxElems = (int64_t)1 << 38;  // grid.x = n / 256 must stay below 2^31

}  // namespace

extern "C" {

int sph2pob_abi_version(void) { return 1; }
const char* sph2pob_target_arch(void) { return "gfx950"; }

const char* sph2pob_error_string(int code) {
    switch (code) {
        case SPH2POB_OK: return "ok";
        case SPH2POB_ERR_NULL: return "null pointer with non-zero element count";
        case SPH2POB_ERR_DIM: return "box_dim must be 4 or 5 (legacy variant: 4 only)";
        case SPH2POB_ERR_OPTION: return "variant/mode/edge/angle/loss option out of range";
        case SPH2POB_ERR_SIZE: return "negative or too large element count";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown sph2pob error";
    }
}

int sph2pob_iou_aligned_f32(const float* b1, const float* b2, float* out, int64_t n, int box_dim, int variant,
                            int mode, int edge, int angle, void* stream) {
    int rc = check_common(box_dim, variant, edge, angle);
    if (rc) return rc;
    if (mode < 0 || mode > 1) return SPH2POB_ERR_OPTION;
    if (n < 0 || n > kMaxElems) return SPH2POB_ERR_SIZE;
    if (n == 0) return SPH2POB_OK;
    if (!b1 || !b2 || !out) return SPH2POB_ERR_NULL;
    return dispatch(variant, box_dim, AlignedLaunch{b1, b2, out, n, mode, edge, angle, (hipStream_t)stream});
}

int sph2pob_iou_pairwise_f32(const float* b1, int64_t m, const float* b2, int64_t n, float* out, int box_dim,
                             int variant, int mode, int edge, int angle, void* stream) {
    int rc = check_common(box_dim, variant, edge, angle);
    if (rc) return rc;
    if (mode < 0 || mode > 1) return SPH2POB_ERR_OPTION;
    if (m < 0 || n < 0 || n > kMaxElems || m > kMaxElems) return SPH2POB_ERR_SIZE;
    if (m == 0 || n == 0) return SPH2POB_OK;
    if (!b1 || !b2 || !out) return SPH2POB_ERR_NULL;
    return dispatch(variant, box_dim, PairwiseLaunch{b1, m, b2, n, out, mode, edge, angle, (hipStream_t)stream});
}

int sph2pob_transform_f32(const float* b1, const float* b2, float* planar1, float* planar2, int64_t n,
                          int box_dim, int variant, int edge, int angle, int jitter, void* stream) {
    int rc = check_common(box_dim, variant, edge, angle);
    if (rc) return rc;
    if (n < 0 || n > kMaxElems) return SPH2POB_ERR_SIZE;
    if (n == 0) return SPH2POB_OK;
    if (!b1 || !b2 || !planar1 || !planar2) return SPH2POB_ERR_NULL;
    return dispatch(variant, box_dim,
                    TransformLaunch{b1, b2, planar1, planar2, n, edge, angle, jitter, (hipStream_t)stream});
}

}  // extern "C"

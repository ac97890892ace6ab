// Shared by the translation units of libsph2pob_hip.so (sph2pob_{iou,assign,loss,nms}.hip): constants, launch knobs, box loads,
// the (VARIANT, DIM) dispatch and the argument checks every entry point starts with.  Internal: include/sph2pob_hip.h is the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/sph2pob_hip.h"
#include "sph2pob_device.hpp"
#include "sph2pob_loss.hpp"
#include "sph2pob_fast.hpp"
#include "sph2pob_unbiased.hpp"

namespace {

using namespace sph2pob;

constexpr int kBlock = 256;  // 4 waves of 64 lanes
constexpr int kCUsDefault = 256;  // MI355X in SPX mode: 8 XCDs x 32 CUs

// tuning / A-B knobs (environment, read once at load): SPH2POB_NO_COMPACT=1 disables the compacting kernels,
// SPH2POB_ALIGNED_KERNEL=persistent selects the persistent form of the aligned kernel for the closed-form arithmetic
// (the default is the one-round chunk form), SPH2POB_NO_PREFETCH=1 its register prefetch, SPH2POB_SLICES_PER_WAVE=s /
// SPH2POB_WGS_PER_CU=k override its grid rule (s slices per wave, or exactly k workgroups per CU), SPH2POB_PW_ROWS the
// pairwise kernel's rows per workgroup
// CU count of the current device (a partitioned MI355X exposes fewer); queried once, no synchronisation involved
static int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            n = v;
        else
            n = kCUsDefault;
    }
    return n;
}
static bool g_no_compact = getenv("SPH2POB_NO_COMPACT") != nullptr;
static bool g_prefetch = getenv("SPH2POB_NO_PREFETCH") == nullptr;
static int g_pw_rows = getenv("SPH2POB_PW_ROWS") ? atoi(getenv("SPH2POB_PW_ROWS")) : 0;
static int g_slices_per_wave = getenv("SPH2POB_SLICES_PER_WAVE") ? atoi(getenv("SPH2POB_SLICES_PER_WAVE")) : 0;
static int g_wgs_per_cu = getenv("SPH2POB_WGS_PER_CU") ? atoi(getenv("SPH2POB_WGS_PER_CU")) : 0;
static bool g_no_prio = getenv("SPH2POB_NO_PRIO") != nullptr;   // A/B: no wave priority in the chunk kernel
static bool g_persistent = getenv("SPH2POB_ALIGNED_KERNEL") != nullptr && getenv("SPH2POB_ALIGNED_KERNEL")[0] == 'p';   // A/B: the persistent form

template <int DIM>
__device__ __forceinline__ void load_box(const float* __restrict__ p, int64_t i, float (&b)[5]) {
    if (DIM == 4) {  // one 16-byte load per lane: 1 KiB per wave instruction, fully coalesced
        // (non-temporal loads: 8.32 vs 8.04 us at 1 M pairs, 51.1 vs 50.4 at 8 M, 97.8 vs 104.1 at 16 M: not kept)
        float4 v = reinterpret_cast<const float4*>(p)[i];
        b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w; b[4] = 0.0f;
    } else {
        const float* q = p + i * 5;
#pragma unroll
        for (int k = 0; k < 5; k++) b[k] = q[k];
    }
}

// FAST: the closed-form core of sph2pob_fast.hpp (standard / efficient with rbb_angle='equator'); otherwise the
// reference-order path of sph2pob_device.hpp (legacy, rbb_angle='project').
template <int VARIANT, int DIM, bool FAST>
__device__ __forceinline__ float pair_iou_sel(const float (&x)[5], const float (&y)[5], int mode, int edge, int angle) {
    if constexpr (VARIANT == VARIANT_UNBIASED) return unbiased_pair_iou<DIM, !FAST>(x, y);
    else if constexpr (VARIANT == VARIANT_NAIVE) return naive_iou<DIM>(x, y, edge == EDGE_TANGENT);   // (edge carries SPH2POB_FLAG_NAIVE_TAN)
    else if constexpr (FAST) return pair_iou_fast<VARIANT, DIM>(x, y, mode, edge);
    else return pair_iou<VARIANT, DIM>(x, y, mode, edge, angle);
}

constexpr int kQCap = 128;                    // per-wave survivor stack capacity (<= 63 carried + 64 pushed)

// number of set bits of a wave mask below this lane: v_mbcnt_lo + v_mbcnt_hi (the 64-bit shift / and / popcount form
// costs eight instructions)
__device__ __forceinline__ int rank_below(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

inline int check_common(int box_dim, int variant_flags, int edge, int angle) {
    const int variant = variant_flags & 0xff;
    if (variant_flags & ~(0xff | SPH2POB_FLAG_REFERENCE_ORDER | SPH2POB_FLAG_ROBUST_PARALLEL | SPH2POB_FLAG_NAIVE_TAN)) return SPH2POB_ERR_OPTION;
    if ((variant_flags & SPH2POB_FLAG_NAIVE_TAN) && variant != SPH2POB_VARIANT_NAIVE) return SPH2POB_ERR_OPTION;
    if (box_dim != 4 && box_dim != 5) return SPH2POB_ERR_DIM;
    if (variant < 0 || variant > SPH2POB_VARIANT_NAIVE || edge < 0 || edge > 2 || angle < 0 || angle > 1) return SPH2POB_ERR_OPTION;
    if (variant >= SPH2POB_VARIANT_LEGACY && variant <= SPH2POB_VARIANT_FOV_IOU && box_dim == 5)
        return SPH2POB_ERR_DIM;  // BFoV-only variants
    return SPH2POB_OK;
}

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SPH2POB_OK : (int)e;
}

// dispatch a (VARIANT, DIM) pair to a functor
template <typename F>
int dispatch(int variant_flags, int box_dim, F&& f) {
    const int variant = variant_flags & 0xff;
    f.fast = !(variant_flags & SPH2POB_FLAG_REFERENCE_ORDER);   // SPH2POB_FLAG_ROBUST_PARALLEL: accepted, always on now
    if (variant == SPH2POB_VARIANT_STANDARD) return box_dim == 4 ? f.template run<0, 4>() : f.template run<0, 5>();
    if (variant == SPH2POB_VARIANT_EFFICIENT) return box_dim == 4 ? f.template run<1, 4>() : f.template run<1, 5>();
    if (variant == SPH2POB_VARIANT_SPH_IOU) return f.template run<3, 4>();
    if (variant == SPH2POB_VARIANT_FOV_IOU) return f.template run<4, 4>();
    if (variant == SPH2POB_VARIANT_UNBIASED) return box_dim == 4 ? f.template run<5, 4>() : f.template run<5, 5>();
    if (variant == SPH2POB_VARIANT_NAIVE) return box_dim == 4 ? f.template run<6, 4>() : f.template run<6, 5>();
    return f.template run<2, 4>();
}

constexpr int64_t kMaxElems = (int64_t)1 << 38;  // grid.x = n / 256 must stay below 2^31

}  // namespace

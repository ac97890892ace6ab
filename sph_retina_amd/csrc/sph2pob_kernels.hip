// libsph2pob_hip.so — kernels + C-ABI launchers (see include/sph2pob_hip.h).  gfx950 only.
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

template <int VARIANT, int DIM, bool FAST>
__global__ __launch_bounds__(kBlock) void iou_aligned_kernel(const float* __restrict__ b1,
                                                            const float* __restrict__ b2,
                                                            float* __restrict__ out, int64_t n, int mode, int edge,
                                                            int angle) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float x[5], y[5];
    load_box<DIM>(b1, i, x);
    load_box<DIM>(b2, i, y);
    out[i] = pair_iou_sel<VARIANT, DIM, FAST>(x, y, mode, edge, angle);
}

// ---- dominant kernel: aligned IoU, closed-form core, with wave-level compaction of the cull survivors ----
// ~60 % of the benchmark distribution's pairs are culled exactly by the bounding-circle test of stage 0 (hardware
// sin/cos, conservative margins); only survivors pay for accurate trig and the clip.  A naive `if (!culled) finish`
// leaves every wave running the finishing stage with ~40 % of its lanes.  Here each wave walks 64-pair slices, pushes
// the survivors' raw boxes on its own LDS stack (ballot + prefix rank => conflict-free consecutive slots, no atomics,
// no barriers: LDS operations of one wave are in order), and runs lean_finish only when 64 records are available, i.e.
// on fully populated waves.  Leftovers of the 4 waves of a workgroup are merged once at the end.
// Slices are grid-strided (slice = wave + k * waves).  A balanced workgroup-contiguous distribution (every workgroup
// floor / ceil of S / G slices instead of 12 next to 8) was measured 10.5 us against 8.9 us at 1 M pairs: at any moment
// the loads of all waves then span the whole 32 MB of input instead of one contiguous ~12 MB window, and the first data
// arrives after 3.0 us instead of 1.85 us (per-wave time stamps, tools/stamp_timeline.py; equal from 2 M pairs up).
// Stores: culled pairs write 0 from stage 0, survivors write from the finishing stage by index.
// Variants built and measured on MI355X this round (1 M pairs; profiles/r02b_ablation_*.log):
//   * a separating-axis reject after stage 1 + a second record stack so that the clip too runs on full waves (only
//     26 % of the pairs overlap, 40 % pass the cull): 8.80 us against 8.55 us without the second stack — the 35 % idle
//     clip lanes cost less than the extra LDS round trip and the longer per-wave dependency chain;
//   * rare lanes (jitter decisions, floors, near-parallel: 0.7 % of the survivors) re-run by a general form in place
//     12.1 us, deferred to an index stack and finished once per workgroup 11.0 us — hence lean_front's guarded blocks;
//   * no carried state at all: one wave per 128-pair chunk, cull both slices, compact, finish (one pass at 81 % lane
//     use, no workgroup merge, no barrier): 8.84 us against 8.96 us here at 1 M pairs, 56.9 against 55.1 us at 8 M;
//     one lane per pair without compaction 10.3 us / 66.0 us (profiles/r02f_ab_*.log);
//   * one survivor ring per WORKGROUP (LDS slot reservation with ds_add_rtn, per-chunk fill counters, chunks claimed by
//     compare-and-swap, slots recycled in order) so that a pass can start as soon as the workgroup holds 64 survivors and
//     all passes but one are full: bit-identical results, 14.2 us / 75.2 us — three dependent LDS round trips and
//     lane-0 sections per slice cost far more than the earlier start gains (profiles/r02i_ab_ring.log).
#if defined(SPH_STAMPS)
// DIAGNOSTIC BUILD ONLY (tools/stamp_timeline.py; never in the shipped library): per-wave time stamps of the dominant
// kernel, s_memrealtime (100 MHz, chip-wide), written to a buffer of their own that nothing else reads.
__device__ unsigned long long* g_stamps = nullptr;
__device__ __forceinline__ void stamp(int wave_global, int k) {
    if ((threadIdx.x & 63) == 0 && g_stamps) g_stamps[(size_t)wave_global * 8 + k] = __builtin_amdgcn_s_memrealtime();
}
// slot 6: where the wave ran (HW_ID | XCC_ID << 32); slot 7: any per-wave figure the caller wants on the timeline
__device__ __forceinline__ void stamp_where(int wave_global) {
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));
    if ((threadIdx.x & 63) == 0 && g_stamps) g_stamps[(size_t)wave_global * 8 + 6] = hw | ((unsigned long long)xcc << 32);
}
__device__ __forceinline__ void stamp_value(int wave_global, unsigned long long v) {
    if ((threadIdx.x & 63) == 0 && g_stamps) g_stamps[(size_t)wave_global * 8 + 7] = v;
}
#define SPH_STAMP(k) stamp(wave_global, k)
#define SPH_STAMP_WHERE() stamp_where(wave_global)
#define SPH_STAMP_VALUE(v) stamp_value(wave_global, v)
#else
#define SPH_STAMP(k)
#define SPH_STAMP_WHERE()
#define SPH_STAMP_VALUE(v)
#endif
constexpr int kQCap = 128;                    // per-wave stack capacity (<= 63 carried + 64 pushed)
template <int DIM>
struct WaveQueue {
    float f[2 * DIM][kQCap];   // raw (theta, phi, alpha, beta[, gamma]) of both boxes
    int idx[kQCap];            // pair index
};
constexpr int queue_lds_bytes(int dim) { return (kBlock / 64) * (2 * dim + 1) * kQCap * 4 + 64; }

template <int DIM>
__device__ __forceinline__ void queue_store(WaveQueue<DIM>& q, int slot, const float (&j1)[5], const float (&j2)[5], int i) {
#pragma unroll
    for (int k = 0; k < DIM; k++) { q.f[k][slot] = j1[k]; q.f[DIM + k][slot] = j2[k]; }
    q.idx[slot] = i;
}
template <int DIM>
__device__ __forceinline__ int queue_load(const WaveQueue<DIM>& q, int slot, float (&j1)[5], float (&j2)[5]) {
#pragma unroll
    for (int k = 0; k < 5; k++) { j1[k] = k < DIM ? q.f[k][slot] : 0.0f; j2[k] = k < DIM ? q.f[DIM + k][slot] : 0.0f; }
    return q.idx[slot];
}
// number of set bits of a wave mask below this lane: v_mbcnt_lo + v_mbcnt_hi (the 64-bit shift / and / popcount form
// costs eight instructions)
__device__ __forceinline__ int rank_below(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The reference-order finish is 15-35 KB of code per copy (ocml's sinf / cosf / acosf / asinf expansions): called, not
// inlined, so that the kernel's two call sites (loop and tail) share one copy and the kernel stays inside the 64 KB
// instruction cache.
template <int VARIANT, int DIM>
__device__ __attribute__((noinline)) float ref_finish(float a0, float a1, float a2, float a3, float a4, float b0, float b1, float b2,
                                                      float b3, float b4, int mode, int edge, int angle) {
    const float u1[5] = {a0, a1, a2, a3, a4}, u2[5] = {b0, b1, b2, b3, b4};
    return pair_iou<VARIANT, DIM>(u1, u2, mode, edge, angle);
}

// ARC: rbb_edge == 'arc' folded at compile time (the chord / tangent forms pull ocml's sinf / tanf argument reduction
// into the cull and the finishing stage: 8 copies of ~100 instructions the common launch never executes)
// REF: finish with the reference-order arithmetic (pair_iou) instead of the closed-form core: the cull is exact for it as
// well (disjoint planar rectangles give exactly 0 in the reference), so `set_arithmetic('reference')`, rbb_angle='project'
// (same planar sizes and positions, other angles: the circles do not change) and sph2pob_legacy (VARIANT 2: chord form of
// the cull) pay their ~3x VALU only for the survivors, on full waves.
template <int VARIANT, int DIM, bool PREFETCH, bool ARC, bool REF = false>
__global__ __launch_bounds__(kBlock, REF ? 4 : (DIM == 4 ? 7 : 5)) void iou_aligned_compact_kernel(const float* __restrict__ b1,
                                                                                      const float* __restrict__ b2,
                                                                                      float* __restrict__ out, int n,
                                                                                      int mode, int edge_arg) {
    __shared__ WaveQueue<DIM> queues[kBlock / 64];
    __shared__ int leftover[kBlock / 64];
    const int edge = ARC ? (int)EDGE_ARC : (edge_arg & 0xff);
    const int angle = REF ? (edge_arg >> 8) & 1 : (int)ANGLE_EQUATOR;   // reference-order finish only: rbb_angle
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    WaveQueue<DIM>& q = queues[wave];
    const int nslices = (n + 63) >> 6;
    const int wave_global = blockIdx.x * (kBlock / 64) + wave, nwaves = gridDim.x * (kBlock / 64);
    int count = 0;  // wave-uniform stack height
    auto finish = [&](const float (&u1)[5], const float (&u2)[5]) -> float {
        if constexpr (REF) return ref_finish<VARIANT, DIM>(u1[0], u1[1], u1[2], u1[3], u1[4], u2[0], u2[1], u2[2], u2[3], u2[4], mode, edge, angle);
        else return lean_finish<VARIANT, DIM>(u1, u2, mode, edge);
    };
    SPH_STAMP(0);
    SPH_STAMP_WHERE();
    // one slice: cull, push the survivors, finish 64 of them when a full wave of records is available
    auto slice = [&](const float (&x)[5], const float (&y)[5], int sl) {
        const int i = sl * 64 + lane;
        bool surv = false;
        if (i < n) {
            if (fast_cull<DIM, VARIANT == VARIANT_LEGACY>(x, y, edge)) out[i] = 0.0f;
            else surv = true;
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(surv);
        if (surv) queue_store<DIM>(q, count + rank_below(m), x, y, i);
        count += __popcll(m);
        if (sl == wave_global) SPH_STAMP(1);   // first slice culled: its data has arrived
        if (count >= 64) {  // wave-uniform
            count -= 64;
            wave_lds_fence();
            float u1[5], u2[5];
            const int j = queue_load<DIM>(q, count + lane, u1, u2);
            out[j] = finish(u1, u2);
            SPH_STAMP(4);   // (last) in-loop pass done
        }
    };
    auto fetch = [&](int sl, float (&x)[5], float (&y)[5]) {
        const int i = sl * 64 + lane;
        if (sl < nslices && i < n) { load_box<DIM>(b1, i, x); load_box<DIM>(b2, i, y); }
    };
    if (PREFETCH) {
        // software prefetch: the next slice's boxes are in flight while this slice is computed (register double buffer;
        // two register sets used alternately instead of the copy were measured slower: the loop body doubles)
        float nx[5] = {0, 0, 0, 0, 0}, ny[5] = {0, 0, 0, 0, 0};
        fetch(wave_global, nx, ny);
        for (int sl = wave_global; sl < nslices; sl += nwaves) {
            float x[5], y[5];
#pragma unroll
            for (int k = 0; k < 5; k++) { x[k] = nx[k]; y[k] = ny[k]; }
            fetch(sl + nwaves, nx, ny);
            slice(x, y, sl);
        }
    } else {
        for (int sl = wave_global; sl < nslices; sl += nwaves) {
            float x[5] = {0, 0, 0, 0, 0}, y[5] = {0, 0, 0, 0, 0};
            fetch(sl, x, y);
            slice(x, y, sl);
        }
    }
    // merge the < 64 leftovers of the four waves and finish them on as few, as full waves as possible
    SPH_STAMP(2);   // loop done
    SPH_STAMP_VALUE((unsigned long long)count);
    if (lane == 0) leftover[wave] = count;
    __syncthreads();
    SPH_STAMP(3);   // workgroup barrier passed
    const int c0 = leftover[0], c1 = leftover[1], c2 = leftover[2], c3 = leftover[3];
    const int total = c0 + c1 + c2 + c3;   // <= 252: at most one chunk per wave
    if (wave * 64 < total) {
        int k = wave * 64 + lane;
        if (k < total) {
            int w = 0;
            if (k >= c0) { k -= c0; w = 1; if (k >= c1) { k -= c1; w = 2; if (k >= c2) { k -= c2; w = 3; } } }
            float u1[5], u2[5];
            const int j = queue_load<DIM>(queues[w], k, u1, u2);
            out[j] = finish(u1, u2);
        }
    }
    SPH_STAMP(5);   // wave done
}

// ---- dominant kernel since round 2 (closed-form arithmetic): the one-round chunk form of the same pipeline ----
// One wave = one chunk of SLICES x 64 consecutive pairs and no carried state: the wave requests its whole chunk up front
// (the whole input is in flight after the first half microsecond), culls it, compacts the survivors on its own LDS stack
// (~51 of 128 for the benchmark distribution) and finishes them in ONE pass (a second one when more than 64 survive).
// No barrier, no merge, 64 VGPRs, so that the 1 954 workgroups of a 1 M-pair launch are all resident at once (8 per CU)
// and no wave ever runs two finishing passes back to back.  What was measured on MI355X (profiles/r02l_*):
//   * the SIMDs are saturated from the arrival of the first data to the end: a finishing pass costs a SIMD ~1 500 cycles
//     = 0.63 us with 8 resident waves (tools/ubench/finish_rate.hip: ~4 cycles per instruction whatever its kind, not
//     the 2 / 4 / 8 of independent instruction streams), a cull ~190, and the launch takes
//     ~3.0 us (kernel boundary + first data) + the VALU time; removing the finishing arithmetic leaves 4.5 us, removing
//     the cull arithmetic saves 0.9 us (r02l_ab_ablation_*.log);
//   * against the persistent form: 8.8 vs 9.0 us at 1 M pairs, 6.0 vs 6.3 at 500 k, 17.1 vs 18.9 at 2 M, 27.9 vs 30.4
//     at 4 M, 52.5 vs 54.3 at 8 M, RBFoV 11.8 vs 12.7 at 1 M; bit-identical results (r02l_ab_chunk_*.log) — the
//     persistent form's leftover merge also ends in partial passes (7 200 passes against 7 813 here, 6 250 if every
//     pass were full) and puts two passes and a barrier on every wave's critical path;
//   * pooling the survivors of 4 / 8 / 16 waves in one workgroup-wide stack (one ds_add_rtn per wave, one LDS-only
//     barrier, chunks assigned or claimed from a counter) so that all passes but one per workgroup are full: 12 % fewer
//     finishing instructions at 8 waves and NO gain (8.8-8.95 us at 1 M, 55-57 at 8 M; r02l_ab_pool_*.log) — the
//     barrier couples waves that sit on different SIMDs; built, measured, removed (and again with the cull phase at a
//     raised wave priority: 7.19 vs 7.29 us at 1 M, 15.5 vs 14.9 at 2 M, 49.5 vs 47.2 at 8 M: r03j_ab_pool_prio_*.log);
//   * workgroups of 1 / 2 / 4 / 8 / 16 independent waves: the same within 1 % from 100 k to 8 M pairs (16: +3 %;
//     r02z_ab_wg*.log); 4 stays;
//   * rotating which wave of the persistent form takes which leftover chunk: the hardware already rotates the
//     wave -> SIMD placement from workgroup to workgroup (tools/ubench/hwid.hip); +0.3 us, removed.
template <int DIM, int SLICES>
struct ChunkQueue {
    float f[2 * DIM][64 * SLICES];
    int idx[64 * SLICES];
};
constexpr int kChunkSlices = 2;
template <int VARIANT, int DIM, bool ARC, int SLICES, int WAVES = kBlock / 64>
__global__ __launch_bounds__(64 * WAVES, DIM == 4 ? 8 : 7) void iou_aligned_chunk_kernel(const float* __restrict__ b1, const float* __restrict__ b2,
                                                                      float* __restrict__ out, int n, int mode, int edge_arg) {
    __shared__ ChunkQueue<DIM, SLICES> queues[WAVES];
    const int edge = ARC ? (int)EDGE_ARC : (edge_arg & 0xff);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    ChunkQueue<DIM, SLICES>& q = queues[wave];
    const int base = (blockIdx.x * WAVES + wave) * (64 * SLICES);
    if (base >= n) return;   // wave-uniform; the kernel has no barrier
    const int wave_global = blockIdx.x * WAVES + wave;
    (void)wave_global;
    SPH_STAMP(0);
    SPH_STAMP_WHERE();
#if defined(SPH_STAMPS)
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime();
#endif
    // Wave priority.  The SIMD's arbiter serves the oldest wave first, so a wave whose boxes arrive late waits for its
    // cheap cull behind the finishing passes of the waves that got theirs early, and its own pass starts that much later:
    // the tail of a launch whose waves are all resident at once.  The load + cull phase runs at priority 1, the finishing
    // pass at 0 (any level above the pass's does the same): 7.76 -> 7.28 us per 1 M pairs, 12.3 -> 11.8 at 1.5 M, 5.21 -> 5.02
    // at 500 k, 11.66 -> 11.17 for 1 M nearby pairs; from 3 M pairs up, where workgroups start as others retire, it costs
    // 1-2.5 % instead (2.6 M: +2.8 %), and RBFoV launches lose 2.5 % at 1 M: the launcher asks for it for BFoV launches of
    // up to two rounds (profiles/r03g_ab_prio*.log).  A priority that falls (or rises) with the wave's progress through
    // its pass, to keep the waves of a SIMD in step (or to retire them one by one): 7.70 / 7.55 against 7.31 / 7.18 us.
    const bool cull_first = (edge_arg & 0x10000) != 0;
    if (cull_first) __builtin_amdgcn_s_setprio(1);
    // BFoV: lanes past the end of the batch load the last pair again (never stored, never stacked): no zero fill of the
    // sixteen registers, no branch around the loads (8.31 -> 8.24 us per 1 M pairs; RBFoV's twenty dword loads were faster
    // behind the branch: 10.5 vs 10.7 us)
    float x[SLICES][5], y[SLICES][5];
#pragma unroll
    for (int s = 0; s < SLICES; s++) {
        const int i = base + s * 64 + lane;
        if (DIM == 4) {
            const int il = i < n ? i : n - 1;
            load_box<DIM>(b1, il, x[s]);
            load_box<DIM>(b2, il, y[s]);
        } else {
#pragma unroll
            for (int k = 0; k < 5; k++) { x[s][k] = 0.0f; y[s][k] = 0.0f; }
            if (i < n) { load_box<DIM>(b1, i, x[s]); load_box<DIM>(b2, i, y[s]); }
        }
    }
    int count = 0;
#pragma unroll
    for (int s = 0; s < SLICES; s++) {
        const int i = base + s * 64 + lane;
        // (masks combined as masks, not through a bool)
#if defined(SPH_ABL_NOCULL)
        const bool culled = ((lane * 2654435761u + s * 40503u + blockIdx.x) >> 7) % 5 >= 2;   // ABLATION: 40 % survive, no cull arithmetic
#else
        const bool culled = fast_cull<DIM, VARIANT == VARIANT_LEGACY>(x[s], y[s], edge);
#endif
        const bool inside = i < n, surv = inside & !culled;
        if (inside & culled) out[i] = 0.0f;   // (non-temporal stores here and below: 8.42 vs 8.30 us at 1 M, 51.4 vs 48.1 at 8 M)
        const unsigned long long m = __builtin_amdgcn_ballot_w64(surv);
        if (surv) {
            const int slot = count + rank_below(m);
#pragma unroll
            for (int k = 0; k < DIM; k++) { q.f[k][slot] = x[s][k]; q.f[DIM + k][slot] = y[s][k]; }
            q.idx[slot] = i;
        }
        count += __popcll(m);
        if (s == 0) SPH_STAMP(1);
    }
    SPH_STAMP(2);
    SPH_STAMP(3);
    SPH_STAMP_VALUE((unsigned long long)count);
    wave_lds_fence();
    if (cull_first) __builtin_amdgcn_s_setprio(0);
    for (int b = 0; b < count; b += 64) {
        const int slot = b + lane;
        if (slot < count) {
            float u1[5], u2[5];
#pragma unroll
            for (int k = 0; k < 5; k++) { u1[k] = k < DIM ? q.f[k][slot] : 0.0f; u2[k] = k < DIM ? q.f[DIM + k][slot] : 0.0f; }
#if defined(SPH_ABL_NOFINISH)
            out[q.idx[slot]] = u1[0] + u2[1] + u1[2] + u2[3] > 1e30f ? 1.0f : 0.5f;   // ABLATION: no finishing arithmetic
#else
            out[q.idx[slot]] = lean_finish<VARIANT, DIM>(u1, u2, mode, edge);
#endif
        }
    }
    SPH_STAMP(5);
#if defined(SPH_STAMPS)
    if (lane == 0 && g_stamps) g_stamps[(size_t)wave_global * 8 + 4] = __builtin_amdgcn_s_memtime() - clk0;   // shader clocks of the wave's life
#endif
}

// ---- pairwise IoU for the assigner call pattern (few rows x many columns), closed-form core ----
// Order-preserving packed keys for max / first-argmax reductions (the assigner): (float bits mapped to an unsigned order) << 32
// | ~index, so that the maximum key is the maximum value and, among equal values, the SMALLEST index — what torch.max(dim)
// returns.  IoUs are >= +0 (the kernels never produce -0) or -1 for ignored columns.
__device__ __forceinline__ unsigned long long pack_max_key(float v, int64_t j) {
    // IoUs are >= 0 (or -1 for ignored columns): map to an order-preserving unsigned key; ties -> smallest index
    unsigned u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned)(0xffffffffu - (unsigned)j);
}
// the fused assigner's row keys keep bit 0 of the low word free for a "this value occurs at more than one column of the
// tile" flag: low = (0x7fffffff - index) << 1 | flag (indices < 2^31)
__device__ __forceinline__ unsigned long long pack_row_key(float v, unsigned j) {
    unsigned u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | ((0x7fffffffu - j) << 1);
}
__device__ __forceinline__ unsigned row_key_index(unsigned long long key) { return 0x7fffffffu - ((unsigned)key >> 1); }
__device__ __forceinline__ float unpack_max_val(unsigned long long key) {
    unsigned u = (unsigned)(key >> 32);
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long w = __shfl_xor(v, o, 64);
        v = w > v ? w : v;
    }
    return v;
}

// keys travel between ranks as SIGNED 64-bit integers (torch.distributed has no unsigned MAX): top bit flipped
__device__ __forceinline__ long long key_to_signed(unsigned long long k) { return (long long)(k ^ 0x8000000000000000ull); }
__device__ __forceinline__ unsigned long long key_from_signed(long long k) { return (unsigned long long)k ^ 0x8000000000000000ull; }

// One thread owns one column box (anchor); a workgroup covers 256 columns x up to 64 rows (GT).  Per-box cull
// quantities are hoisted: rows live in LDS (broadcast reads), the column's in registers, so a culled pair costs
// ~15 VALU instructions + one coalesced store of 0.  Survivors are (row, column) index pairs pushed on the wave's
// LDS stack and finished 64 at a time on fully populated waves (same scheme as iou_aligned_compact_kernel).
constexpr int kPwRows = 64;
// ARC: rbb_edge == 'arc' folded at compile time, as in the aligned kernels (with a run-time edge the chord / tangent forms
// made this kernel 47 KB of code at 84 VGPRs).  The < 64 leftovers of the four waves are merged once at the end and
// finished on as few, as full waves as possible (with 8 rows per workgroup a wave stacks ~80 survivors: one full pass and
// a 16-lane one without the merge).
// OUT: bit 0 = write the m x n matrix; bit 1 = the assigner's reductions (max_iou_assigner.py:171-176 without
// the matrix) — every finished survivor with IoU > 0 goes into the tile's per-column and per-row maxima in LDS (ds_max_u64 on
// packed keys; culled pairs and survivors that finish to 0 are covered by the initial values: exact zeros), written once per
// workgroup as
//   col_part[chunk][j]      max over the chunk's rows of (IoU[i][j], first row)         (chunk = blockIdx row chunk)
//   row_part[i][tile]       max over the tile's 256 columns of (IoU[i][j], first column, global index = col_offset + j),
//                           bit 0 = the value may occur at more than one column of the tile
//   row_acc[i]              atomic max of the row's partials over ALL tiles (zero before the launch)
// `ignore` (optional, one byte per column): columns whose overlaps the assigner sets to -1 (max_iou_assigner.py:115-126)
// — they take part in no row maximum, their column maximum is (-1, row 0), and the matrix, when written, holds -1.
// Per-GT accumulators of the fused assigner (the "state" buffer, zero between calls): one u64 per GT, 256 bytes apart for
// k <= 1024 so that the GTs' atomics spread over memory channels (64 x 392 832 anchors: 70.2 -> 65.5 us, profiles/r05d_ab_fused_stride.log;
// 16 slots per GT instead made the finalize pass read 64 KB per workgroup: slower, profiles/r05e_ab_fused_slots.log), followed by the
// finalize pass's arrival counters: one per group of 32 column tiles + one for the groups, each on a 64-byte line (a single
// counter serialises one returning atomic per workgroup: 1 535 tiles = +13 us, profiles/r05b_trace_fused_one_counter.txt).
constexpr int kAccLine = 8;        // u64 words per counter line
constexpr int kTicketGroup = 32;   // column tiles per first-level arrival counter
__host__ __device__ inline int64_t acc_stride(int64_t k) { return k <= 1024 ? 32 : 1; }
__host__ __device__ inline int64_t acc_words(int64_t k) { return (k * acc_stride(k) + kAccLine - 1) / kAccLine * kAccLine; }
__host__ __device__ inline int64_t ticket_groups(int64_t tiles) { return (tiles + kTicketGroup - 1) / kTicketGroup; }
constexpr int kRowSlots = 4;   // LDS copies of a row's running maximum (lane & 3): a pass holds a few rows, 64 lanes on one address serialise
template <int VARIANT, int DIM, bool ARC, int OUT = 1>
__global__ __launch_bounds__(kBlock, ARC ? 8 : 4) void iou_pairwise_compact_kernel(const float* __restrict__ b1, int m,
                                                                     const float* __restrict__ b2, int n,
                                                                     float* __restrict__ out, int mode, int edge_arg,
                                                                     int rows_per_wg,
                                                                     const unsigned char* __restrict__ ignore = nullptr,
                                                                     unsigned long long* __restrict__ col_part = nullptr,
                                                                     unsigned long long* __restrict__ row_part = nullptr,
                                                                     unsigned col_offset = 0,
                                                                     unsigned long long* __restrict__ row_acc = nullptr) {
    constexpr bool MATRIX = (OUT & 1) != 0, REDUCE = (OUT & 2) != 0;
    __shared__ float row_raw[kPwRows][5];
    __shared__ float4 row_cull[kPwRows];
    __shared__ int2 stack[kBlock / 64][kQCap];
    __shared__ int leftover[kBlock / 64];
    // sin / cos of every box's jittered colatitude, once per box instead of once per surviving pair (two of the three
    // sincos of a finishing pass): rows by their first threads, columns by their owner; the finishing lane — any lane,
    // the survivors are compacted — reads them by index
    __shared__ ColatTrig row_trig[kPwRows];
    __shared__ ColatTrig col_trig[kBlock];
    __shared__ float col_raw[DIM][kBlock];   // the finishing lanes read the column's box here, not from global memory (bit-equal;
                                             // fused 64 x 98 208: 31.2 -> 30.5 us, 64 x 392 832: 57.4 -> 53.6: profiles/r05a_ab_fused_variants.log)
    __shared__ int row_tie[REDUCE ? kPwRows : 1];
    __shared__ unsigned long long col_key[REDUCE ? kBlock : 1];
    __shared__ unsigned long long row_key[REDUCE ? kPwRows : 1][kRowSlots];
    __shared__ unsigned long long tile_base;   // what a row holds before any survivor: (0, first live column) or (-1, first ignored one)
    const int edge = ARC ? (int)EDGE_ARC : edge_arg;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // dispatch order = LAST column tile first, all of its row chunks, then the tile before it: anchor grids end with their
    // coarsest level (mmdet's AnchorGenerator walks the strides upwards) and their tiles grow heavier towards the end —
    // the coarsest anchors survive the cull against nearly every GT and carry the longest serial chains of passes —, and
    // the grid is larger than what is resident at once: dispatched last, the heaviest tiles started last.  Heaviest first:
    // 64 x 98 208 anchors 21.7 -> 18.5 us, 64 x 392 832 47.2 -> 41.3 us with the rows-per-workgroup rule retuned for it
    // (profiles/r03y_ab_pairwise*.log, r03z_ab_pairwise.log); tiles taken from both ends inwards instead: 20.2 / 42.1 us.
    // A caller that lists the coarse level first gets the previous behaviour.  (tiles x chunks <= m n / 1024 + ..., and the
    // m x n matrix has to fit the device: the linear id stays far below 2^32.  The same order from a transposed grid —
    // chunks on x, tiles on y, no division — measured 1 % slower at 392 832 anchors: r04b_ab_pairwise_transposed.log.)
    const unsigned lid = blockIdx.y * gridDim.x + blockIdx.x;
    const int bx = (int)(gridDim.x - 1 - lid / gridDim.y), by = (int)(lid % gridDim.y);
    const int r0 = by * rows_per_wg, rows = (m - r0 < rows_per_wg) ? m - r0 : rows_per_wg;
    if ((int)threadIdx.x < rows) {
        float g[5];
        load_box<DIM>(b1, r0 + threadIdx.x, g);
        CullBox cg = cull_box(g, edge);
#pragma unroll
        for (int k = 0; k < 5; k++) row_raw[threadIdx.x][k] = g[k];
        row_cull[threadIdx.x] = make_float4(cg.s, cg.c, cg.th_rev, cg.r);
        row_trig[threadIdx.x] = colat_trig(g[1], 1);
    }
    const int j = bx * kBlock + threadIdx.x;
    const bool valid = j < n;
    float a[5] = {0.0f, 0.0f, 1.0f, 1.0f, 0.0f};
    if (valid) load_box<DIM>(b2, j, a);
    const bool ign = (REDUCE || MATRIX) && ignore != nullptr && valid && ignore[j] != 0;
    const bool live = valid & !ign;
    col_trig[threadIdx.x] = colat_trig(a[1], 2);
#pragma unroll
    for (int k = 0; k < DIM; k++) col_raw[k][threadIdx.x] = a[k];
    if constexpr (REDUCE) {
        col_key[threadIdx.x] = pack_max_key(ign ? -1.0f : 0.0f, r0);
        if (threadIdx.x < kPwRows * kRowSlots) (&row_key[0][0])[threadIdx.x] = 0ull;
        if (kPwRows * kRowSlots > kBlock && threadIdx.x + kBlock < kPwRows * kRowSlots) (&row_key[0][0])[threadIdx.x + kBlock] = 0ull;
        if (threadIdx.x == 0) tile_base = 0ull;
        if (threadIdx.x < kPwRows) row_tie[threadIdx.x] = 0;
    }
    __syncthreads();
    if constexpr (REDUCE) {   // first live / first ignored column of each wave -> the tile's base key (a max over <= 8 candidates)
        const unsigned long long ml = __builtin_amdgcn_ballot_w64(live), mi = __builtin_amdgcn_ballot_w64(ign);
        if (lane == 0) {
            const int jw = bx * kBlock + wave * 64;
            if (ml) atomicMax(&tile_base, pack_row_key(0.0f, col_offset + jw + __builtin_ctzll(ml)));
            if (mi) atomicMax(&tile_base, pack_row_key(-1.0f, col_offset + jw + __builtin_ctzll(mi)));
        }
    }
    const CullBox ca = cull_box(a, edge);
    int2* st = stack[wave];
    int count = 0;
    auto finish_one = [&](int2 e) {
        float g[5], p[5];
#pragma unroll
        for (int k = 0; k < 5; k++) g[k] = row_raw[e.x][k];
#pragma unroll
        for (int k = 0; k < 5; k++) p[k] = k < DIM ? col_raw[k < DIM ? k : 0][e.y - bx * kBlock] : 0.0f;
        const float v = lean_finish<VARIANT, DIM, 1>(g, p, mode, edge, row_trig[e.x], col_trig[e.y - bx * kBlock]);
        if constexpr (MATRIX) out[(int64_t)(r0 + e.x) * n + e.y] = v;
        if constexpr (REDUCE) if (!(v <= 0.0f)) {   // > 0 or NaN: zeros are the initial values
            atomicMax(&col_key[e.y - bx * kBlock], pack_max_key(v, r0 + e.x));
            // the value already in the slot (a column of this row finished earlier) — equal value bits = a tie inside the tile
            const unsigned long long key = pack_row_key(v, col_offset + e.y);
            const unsigned long long old = atomicMax(&row_key[e.x][lane & (kRowSlots - 1)], key);
            if ((unsigned)(old >> 32) == (unsigned)(key >> 32)) row_tie[e.x] = 1;
        }
    };
    float* orow = out + (int64_t)r0 * n + j;   // this column's element of the tile's first row (cull rows at a higher
                                               // wave priority than the passes, as in the chunk kernel: no gain here)
    for (int i = 0; i < rows; i++) {
        const float4 rc = row_cull[i];
        const bool culled = cull_pair(CullBox{rc.x, rc.y, rc.z, rc.w}, ca), surv = live & !culled;
        if constexpr (MATRIX) {
            if (valid & (culled | ign)) *orow = ign ? -1.0f : 0.0f;
            orow += n;
        }
        const unsigned long long mk = __builtin_amdgcn_ballot_w64(surv);
        if (surv) st[count + rank_below(mk)] = make_int2(i, j);
        count += __popcll(mk);
        if (count >= 64) {
            count -= 64;
            wave_lds_fence();
            finish_one(st[count + lane]);
        }
    }
    // merge the < 64 leftovers of the four waves
    if (lane == 0) leftover[wave] = count;
    // LDS only: __syncthreads() would also wait for the acknowledgement of every store above (vmcnt(0))
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int c0 = leftover[0], c1 = leftover[1], c2 = leftover[2], c3 = leftover[3];
    const int total = c0 + c1 + c2 + c3;   // <= 252: at most one chunk per wave
    if (wave * 64 < total) {
        int k = wave * 64 + lane;
        if (k < total) {
            int w = 0;
            if (k >= c0) { k -= c0; w = 1; if (k >= c1) { k -= c1; w = 2; if (k >= c2) { k -= c2; w = 3; } } }
            finish_one(stack[w][k]);
        }
    }
    if constexpr (REDUCE) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (valid) col_part[(int64_t)by * n + j] = col_key[threadIdx.x];
        if ((int)threadIdx.x < rows) {
            unsigned long long best = tile_base;
#pragma unroll
            for (int t = 0; t < kRowSlots; t++) { const unsigned long long v = row_key[threadIdx.x][t]; best = v > best ? v : best; }
            int same = 0;   // slots that hold the maximum VALUE (each at a column of its own)
#pragma unroll
            for (int t = 0; t < kRowSlots; t++) same += (unsigned)(row_key[threadIdx.x][t] >> 32) == (unsigned)(best >> 32);
            // flag: conservative (a tie seen at a lower value also sets it; the finalize pass then only re-evaluates for nothing)
            row_part[(int64_t)(r0 + threadIdx.x) * gridDim.x + bx] = best | (unsigned long long)(row_tie[threadIdx.x] | (same > 1));
            // the row's running maximum over all tiles: relaxed device-scope atomic, only from tiles that hold something
            // above the rows' common floor (0 at the shard's first column) — with an ignore mask the floor is not known
            // here, and every tile contributes
            if (ignore != nullptr || (unsigned)(best >> 32) > 0x80000000u)
                __hip_atomic_fetch_max(row_acc + (r0 + threadIdx.x) * acc_stride(m), best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// out[i*n + j]: consecutive lanes walk j (coalesced stores, b2 loads coalesced, b1 row is a broadcast).
template <int VARIANT, int DIM, bool FAST>
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
    out[i * n + j] = pair_iou_sel<VARIANT, DIM, FAST>(x, y, mode, edge, angle);
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


// ---- planar rotated IoU on given planar boxes (mmcv box_iou_rotated / diff_iou_rotated_2d values) ----
__global__ __launch_bounds__(kBlock) void planar_iou_kernel(const float* __restrict__ p1, int64_t m, const float* __restrict__ p2,
                                                           int64_t n, float* __restrict__ out, int aligned, int mode) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t i = aligned ? j : (int64_t)blockIdx.y;
    if (j >= n) return;
    const float* a = p1 + i * 5;
    const float* b = p2 + j * 5;
    const PBox A{a[0], a[1], a[2], a[3], a[4]}, B{b[0], b[1], b[2], b[3], b[4]};
    out[aligned ? j : i * n + j] = planar_iou(A, B, mode);
}

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

// ---- NMS (replaces the python greedy loop of sph_nms_op, sphdet/bbox/nms/sph_nms.py:62-74) ----
// Boxes arrive sorted by (class, descending score).  Kernel 1: one wave per (row i, 64-column word w) evaluates
// IoU(box_i [role bboxes1], box_j [role bboxes2]) > thr for the 64 columns j = 64w + lane (j > i, same class)
// and emits the 64-bit suppression word with one ballot — no LDS, no atomics.  Kernel 2: a single wave sweeps
// the rows in order; the greedy dependency inside a 64-row block is resolved on the 64x64 diagonal block held
// one row per lane (readlane, scalar bit ops), then the kept rows of the block are OR-ed into the running
// "removed" bit-vector (LDS) with lanes striding over the words, so global loads are never on the serial chain.
// A row's class segment [begin, end) in the class-sorted order, found by the whole wave: 64 probes per round, both ends in
// the same rounds — three dependent round trips for 5 000 rows where a per-lane binary search made 2 x 13 (the NMS kernels
// of a 5 000-box call spent most of their ~12 us in those searches).  All lanes of the wave pass the same i.
__device__ __forceinline__ void wave_class_segment(const int64_t* __restrict__ cls, int64_t k, int64_t i, int64_t& seg_begin, int64_t& seg_end,
                                                   bool want_begin = true) {
    const int lane = threadIdx.x & 63;
    const int64_t ci = cls[i];
    // end: the first e in (i, k] with e == k or cls[e] > ci; invariant: cls[elo - 1] <= ci, answer in [elo, ehi]
    // begin: the first b in [0, i] with cls[b] >= ci (== ci); invariant: answer in [blo, bhi], cls[bhi] >= ci
    int64_t elo = i + 1, ehi = k, blo = 0, bhi = want_begin ? i : 0;
    if (!want_begin) blo = 0;
    while (elo < ehi || blo < bhi) {
        const int64_t es = (ehi - elo + 63) / 64, bs = (bhi - blo + 63) / 64;
        const int64_t ep = elo + lane * es, bp = blo + lane * bs;
        const bool ein = elo < ehi && ep < ehi, bin = blo < bhi && bp < bhi;
        const int64_t ce = cls[ein ? ep : i], cb = cls[bin ? bp : i];   // (two independent loads per round)
        const int te = __popcll(__builtin_amdgcn_ballot_w64(ein && ce <= ci));   // probes still inside the class (monotone)
        const int tb = __popcll(__builtin_amdgcn_ballot_w64(bin && cb < ci));    // probes still in front of it
        if (elo < ehi) {
            const int64_t base = elo;
            if (te == 0) ehi = elo;
            else { elo = base + (te - 1) * es + 1; const int64_t cap = base + te * es; ehi = cap < ehi ? cap : ehi; }
        }
        if (blo < bhi) {
            const int64_t base = blo;
            if (tb == 0) bhi = blo;
            else { blo = base + (tb - 1) * bs + 1; const int64_t cap = base + tb * bs; bhi = cap < bhi ? cap : bhi; }
        }
    }
    seg_end = elo;
    seg_begin = want_begin ? blo : 0;
}

template <int VARIANT, int DIM, bool FAST>
__global__ __launch_bounds__(kBlock) void nms_mask_kernel(const float* __restrict__ boxes,
                                                         const int64_t* __restrict__ cls, int64_t k, int words,
                                                         float thr, unsigned long long* __restrict__ mask, int edge) {
    // one wave per row i: only the words that hold later columns of row i's own class segment are evaluated.
    // Row layout: `words` u64 per row, word r of row i covers columns 64 * ((seg_start >> 6) + r) ...: indices are
    // relative to the row's class segment, so the matrix is k x (largest segment / 64 + 2) instead of k x k / 64.
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + wave;
    if (i >= k) return;
    int64_t seg_begin = 0, seg_end = k;  // row i's class segment [seg_begin, seg_end) (boxes are sorted by class)
    if (cls) wave_class_segment(cls, k, i, seg_begin, seg_end);
    const int64_t base = seg_begin >> 6;
    // relative word range of the later columns; empty when seg_end == i + 1; clipped to the row (a caller that
    // under-states the largest segment gets truncated suppression, never an out-of-bounds store)
    const int64_t r_first = ((i + 1) >> 6) - base;
    int64_t r_last = ((seg_end - 1) >> 6) - base;
    if (r_last > words - 1) r_last = words - 1;
    const bool none = seg_end <= i + 1;
    unsigned long long* row = mask + i * words;
    for (int w = lane; w < words; w += 64)
        if (none || w < r_first || w > r_last) row[w] = 0ull;
    if (none) return;
    float x[5];
    load_box<DIM>(boxes, i, x);
    for (int64_t r = r_first; r <= r_last; r++) {
        const int64_t j = (base + r) * 64 + lane;
        bool hit = false;
        if (j > i && j < seg_end) {
            float y[5];
            load_box<DIM>(boxes, j, y);
            hit = pair_iou_sel<VARIANT, DIM, FAST>(x, y, MODE_IOU, edge, ANGLE_EQUATOR) > thr;
        }
        unsigned long long bits = __builtin_amdgcn_ballot_w64(hit);
        if (lane == 0) row[r] = bits;
    }
}

constexpr int kNmsMaxWords = 512;  // <= 32768 boxes per class segment (the sweep's removed bit-vector lives in LDS)

// The same suppression matrix for the closed-form variants (sph2pob_standard / sph2pob_efficient), with the cull and the
// wave-level compaction of the IoU kernels: a row's later columns are culled with the row's bounding-circle quantities
// hoisted (most same-class candidates of a detector are far apart), the survivors' column indices go on the wave's LDS
// stack, lean_finish runs on 64 of them at a time, and a hit sets its bit in the row's bitmap in LDS (ds_or), which is
// written out once.  50 M candidate pairs of the 61 k-candidate pipeline scene: 465 us with one lane per pair.
template <int VARIANT, int DIM>
__global__ __launch_bounds__(kBlock) void nms_mask_compact_kernel(const float* __restrict__ boxes, const int64_t* __restrict__ cls,
                                                                 int64_t k, int words, float thr,
                                                                 unsigned long long* __restrict__ mask) {
    __shared__ int stack[kBlock / 64][kQCap];
    __shared__ unsigned int bits[kBlock / 64][2 * kNmsMaxWords];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + wave;
    if (i >= k) return;   // whole waves leave: no workgroup barrier below
    int64_t seg_begin = 0, seg_end = k;
    if (cls) wave_class_segment(cls, k, i, seg_begin, seg_end);
    const int64_t base = seg_begin >> 6;
    const int64_t r_first = ((i + 1) >> 6) - base;
    int64_t r_last = ((seg_end - 1) >> 6) - base;
    if (r_last > words - 1) r_last = words - 1;
    const bool none = seg_end <= i + 1;
    unsigned int* bm = bits[wave];
    for (int w = lane; w < 2 * words; w += 64) bm[w] = 0u;
    unsigned long long* row = mask + i * words;
    if (!none) {
        float x[5];
        load_box<DIM>(boxes, i, x);
        const CullBox cx = cull_box(x, EDGE_ARC);
        const ColatTrig xt = colat_trig(x[1], 1);   // the row's colatitude trig: once per row, not once per surviving pair
        int* st = stack[wave];
        int count = 0;
        auto finish_one = [&](int j) {
            float y[5];
            load_box<DIM>(boxes, j, y);
            if (lean_finish<VARIANT, DIM, 2>(x, y, MODE_IOU, EDGE_ARC, xt) > thr) {
                const int rel = j - (int)(base << 6);
                atomicOr(&bm[rel >> 5], 1u << (rel & 31));
            }
        };
        wave_lds_fence();   // the bitmap is zero before the first hit
        for (int64_t r = r_first; r <= r_last; r++) {
            const int64_t j = (base + r) * 64 + lane;
            bool surv = false;
            if (j > i && j < seg_end) {
                float y[5];
                load_box<DIM>(boxes, j, y);
                surv = !cull_pair(cx, cull_box(y, EDGE_ARC));
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(surv);
            if (surv) st[count + rank_below(m)] = (int)j;
            count += __popcll(m);
            if (count >= 64) {
                count -= 64;
                wave_lds_fence();
                finish_one(st[count + lane]);
            }
        }
        wave_lds_fence();
        if (lane < count) finish_one(st[lane]);
        wave_lds_fence();   // every ds_or of this wave has been issued before the bitmap is read back (in-order LDS)
    }
    for (int w = lane; w < words; w += 64) row[w] = (unsigned long long)bm[2 * w] | ((unsigned long long)bm[2 * w + 1] << 32);
}

// Greedy sweep, one WORKGROUP per class segment (classes are independent).  Every workgroup looks at 4 candidate rows;
// a row that starts a class segment makes the whole workgroup sweep that segment's 64-row blocks in order, as a two-stage
// pipeline with one barrier per block:
//   wave 0 resolves the serial dependency inside block b on the 64x64 diagonal block held one row per lane (v_readlane
//   + scalar bit operations only), publishes the kept rows, and takes the kept rows' word b + 1 — the only word the NEXT
//   resolve needs from this block — from a register it loaded one block ahead, with the diagonal;
//   the other seven waves meanwhile OR the kept rows of block b - 1 into the words from b + 1 on of the segment's
//   "removed" bit-vector (LDS): a task is (word, group of 16 rows) — 16 independent loads in flight per thread —
//   combined with an LDS atomic OR.  Word b of the bit-vector is complete when block b is resolved: blocks up to b - 2
//   reached it through the OR stage (a barrier ago at least), block b - 1 through wave 0's register.
// (All threads ORing after every resolve, two barriers per block: 1.85 us per block; one wave per segment: 2.4 us.  An OR
// stage two blocks deep — wave 0 carrying words b + 1 and b + 2 in registers, the workers' loads left in flight across an
// LDS-only barrier — was built and measured: 91 us against this form's 83 for one class of 5 000, 15.4 against 12.8 us
// for 37 classes; the compiler waits for the loads at the loop's register copies anyway.  Not kept.)
constexpr int kSweepBlock = 512, kSweepCands = 4;
__global__ __launch_bounds__(kSweepBlock) void nms_sweep_kernel(const unsigned long long* __restrict__ mask,
                                                                const int64_t* __restrict__ cls, int64_t k, int words,
                                                                unsigned char* __restrict__ keep) {
    __shared__ unsigned long long removed[kNmsMaxWords];
    __shared__ unsigned long long kept_sh[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int cand = 0; cand < kSweepCands; cand++) {   // workgroup-uniform loop and conditions: barriers are safe
        const int64_t s = (int64_t)blockIdx.x * kSweepCands + cand;  // candidate segment head
        if (s >= k) break;
        if (cls ? (s > 0 && cls[s] == cls[s - 1]) : (s > 0)) continue;
        int64_t seg_end = k;
        if (cls) { int64_t unused; wave_class_segment(cls, k, s, unused, seg_end, false); }
        // blocks of 64 rows, numbered relative to the segment's first block (the mask rows use the same numbering)
        const int64_t base = s >> 6;
        int b_last = (int)(((seg_end - 1) >> 6) - base);
        const int limit = (words < kNmsMaxWords ? words : kNmsMaxWords) - 1;
        if (b_last > limit) b_last = limit;   // over-long segment: rows beyond the limit keep 0 flags
        __syncthreads();                      // the previous candidate's sweep is done with the shared state
        for (int w = threadIdx.x; w <= b_last; w += kSweepBlock) removed[w] = 0ull;
        for (int64_t r = (base + b_last + 1) * 64 + threadIdx.x; r < seg_end; r += kSweepBlock) keep[r] = 0;
        __syncthreads();
        // the diagonal word of a block's rows and the word after it depend on nothing the sweep computes: the next
        // block's are requested while this block is resolved (the loads' latency was on every block's critical path)
        auto load_word = [&](int b, int w) -> unsigned long long {
            const int64_t row = (base + b) * 64 + lane;
            return (wave == 0 && b <= b_last && w <= b_last && row >= s && row < seg_end) ? mask[row * words + w] : 0ull;
        };
        unsigned long long diag_next = load_word(0, 0), after_next = load_word(0, 1);
        unsigned long long carry = 0ull;   // wave 0, uniform: what block b - 1's kept rows remove in word b
        for (int b = 0; b <= b_last; b++) {
            if (wave == 0) {
                const int64_t row = (base + b) * 64 + lane;
                const bool mine = row >= s && row < seg_end;
                const unsigned long long diag = diag_next, after = after_next;
                diag_next = load_word(b + 1, b + 1);
                after_next = load_word(b + 1, b + 2);
                const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
                const unsigned alo = (unsigned)after, ahi = (unsigned)(after >> 32);
                // scalar (SGPR) state of the serial chain; rows of the block outside this segment start out "removed"
                // (the readlane builtins return int: cast before widening, or bit 31 sign-extends)
                unsigned long long rem = removed[b] | carry | ~__builtin_amdgcn_ballot_w64(mine);
                rem = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(rem >> 32)) << 32) |
                      (unsigned)__builtin_amdgcn_readfirstlane((unsigned)rem);
                // one step per KEPT row (s_ff1 on the rows still alive), not per row: a dense scene keeps a few of 64
                unsigned long long keepbits = 0ull;
                carry = 0ull;
                while (~rem != 0ull) {
                    const int r = __builtin_ctzll(~rem);
                    keepbits |= 1ull << r;
                    rem |= (1ull << r) | ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(dhi, r) << 32) |
                           (unsigned)__builtin_amdgcn_readlane(dlo, r);
                    carry |= ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(ahi, r) << 32) |
                             (unsigned)__builtin_amdgcn_readlane(alo, r);
                }
                if (mine) keep[row] = (unsigned char)((keepbits >> lane) & 1ull);
                if (lane == 0) kept_sh[b & 1] = keepbits;
            } else if (b >= 1) {
                // OR stage for block b - 1 (its kept rows were published before the last barrier): words b + 1 ... b_last
                const unsigned long long keepbits = kept_sh[(b - 1) & 1];
                const int64_t row0 = (base + b - 1) * 64;
                const int ntasks = (b_last - b) * 4;   // (word, group of 16 rows)
                for (int t = threadIdx.x - 64; t < ntasks; t += kSweepBlock - 64) {
                    const int w = b + 1 + (t >> 2), r0 = (t & 3) * 16;
                    const unsigned bits = (unsigned)(keepbits >> r0) & 0xffffu;
                    if (bits == 0u) continue;
                    const unsigned long long* col = mask + (row0 + r0) * words + w;
                    unsigned long long v[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) v[u] = ((bits >> u) & 1u) ? col[(int64_t)u * words] : 0ull;
                    unsigned long long acc = 0ull;
#pragma unroll
                    for (int u = 0; u < 16; u++) acc |= v[u];
                    if (acc) atomicOr(&removed[w], acc);
                }
            }
            __syncthreads();
        }
    }
}

// ---- batched NMS without the host (sph_batched_nms, sphdet/bbox/nms/sph_nms.py:22-60, for K <= 16 384 candidates) ----
// The reference sorts per class on the host and loops; round 2 sorted with two stable torch sorts (2 x ~30 us of rocPRIM
// passes at K = 5 000), a scatter, a masked gather and their launch gaps: 0.18 ms around 26 us of kernels.  Here:
//   nms_prepare_kernel   composite 64-bit keys (class | descending score | index), unique by construction, and a RANK sort:
//                        position of a box = number of smaller keys — K^2 compares (25 M at K = 5 000) spread over the whole
//                        chip, every workgroup holding all keys in LDS (64 boxes x 8 key ranges per workgroup; a one-workgroup
//                        bitonic network in LDS was built first: LDS-bandwidth-bound, ~70 us per sort) -> boxes / classes in
//                        (class, -score) order, the permutation and the (descending score | index) keys in that order;
//   mask + sweep         as before, on the full k x (k / 64 + 2) layout (no segment width needed from the host);
//   nms_select_kernel    the same rank sort among the KEPT boxes by (descending score | index) -> the first max_num kept indices
//                        in the reference's final order (:49-52), dets = (box, score), and their count.
// Ties are broken by the original index (stable), as in round 2.  One host read (the count) sizes the outputs.
__device__ __forceinline__ unsigned desc_score_bits(float v) {   // larger score -> smaller unsigned
    unsigned u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~u;
}
constexpr int kNmsIdxBits = 14, kNmsClsBits = 18;   // K <= 16 384 candidates, class ids in [0, 262 143]
__device__ __forceinline__ unsigned long long nms_class_key(int64_t c, float score, int j) {
    return ((unsigned long long)(c & (((int64_t)1 << kNmsClsBits) - 1)) << (32 + kNmsIdxBits)) |
           ((unsigned long long)desc_score_bits(score) << kNmsIdxBits) | (unsigned)j;
}
// Rank of IPW keys among all keys, the keys held in REGISTERS: wave w of the workgroup holds the slice [w * T * 64, (w + 1) * T * 64)
// of the key sequence, one key per lane and register (coalesced loads, no LDS); the key whose rank is wanted is wave-uniform
// (v_readlane -> SGPR pair), one v_cmp_lt_u64 tests it against 64 keys and s_bcnt1 counts.  (First form: all keys in LDS, the
// wanted keys one per lane, the others read as LDS broadcasts — one LDS instruction per 64 compares made the CU's single LDS
// the limiter: 12.4 us per launch at K = 5 000; a one-workgroup bitonic network in LDS before that: ~70 us.)
// `mine`: lane a < IPW holds the key of the workgroup's a-th box.  Returns, in lanes a < IPW of EVERY wave, the number of keys
// below it; `part` is (BS / 64) x 64 ints of LDS.
// (Every slice register takes part in every test — the padding keys are ~0, never below anything: with a per-register
// `t < tcount` skip each test sat behind its own scalar branch and the v_cmp -> s_bcnt1 pairs ran one by one, 6 us per launch.)
template <int T, int IPW, int BS>
__device__ __forceinline__ int rank_against_slices(const unsigned long long (&key)[T], unsigned long long mine, int (*part)[64]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned mlo = (unsigned)mine, mhi = (unsigned)(mine >> 32);
    // per-lane counters, one per wanted key: v_cmp + v_addc per test, no scalar round trip (popcount of each compare's mask
    // on the scalar unit cost ~75 cycles per test: 0.6 us per slice register); the lanes are added up once at the end
    unsigned long long ka[IPW];
#pragma unroll
    for (int a = 0; a < IPW; a++)
        ka[a] = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mhi, a) << 32) | (unsigned)__builtin_amdgcn_readlane((int)mlo, a);
    int cnt[IPW];
#pragma unroll
    for (int a = 0; a < IPW; a++) cnt[a] = 0;
    // (the borrow of key - ka added with carry — three full-rate instructions in place of v_cmp_lt_u64 + v_cndmask + v_add — was
    // measured too: not faster; what sets the time is how many workgroups a CU has to run one after the other, see the launcher)
#pragma unroll
    for (int t = 0; t < T; t++) {
#pragma unroll
        for (int a = 0; a < IPW; a++) cnt[a] += key[t] < ka[a] ? 1 : 0;
    }
    int mycnt = 0;
#pragma unroll
    for (int a = 0; a < IPW; a++) {
        int v = cnt[a];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        mycnt = lane == a ? v : mycnt;
    }
    part[wave][lane] = mycnt;
    __syncthreads();
    int r = 0;
#pragma unroll
    for (int w = 0; w < BS / 64; w++) r += part[w][lane];
    return r;
}
template <int T, int DIM, int IPW, int BS>
__global__ __launch_bounds__(BS) void nms_prepare_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                                const int64_t* __restrict__ idxs, int k,
                                                                float* __restrict__ boxes_sorted, int64_t* __restrict__ cls_sorted,
                                                                int* __restrict__ order, unsigned long long* __restrict__ skey_sorted,
                                                                int* __restrict__ status) {
    __shared__ int part[BS / 64][64];   // (T keys per lane)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tcount = ((k + BS - 1) / BS);   // registers in use: the slices cover [0, tcount * BS)
    unsigned long long key[T];
    int bad = 0;
    // every load of the kernel — the slice, this workgroup's own boxes' scores / classes AND their coordinates (the gather at
    // the end depends on i only, not on the rank) — is requested here, before the first key is built: one round trip to memory
    // instead of three (first form: 11 us per launch, latency-bound)
    const int i = blockIdx.x * IPW + lane, ic = i < k ? i : k - 1;
    const float si = scores[ic];
    const int64_t ci = idxs ? idxs[ic] : 0;
    float bx[5];
    load_box<DIM>(boxes, ic, bx);
    {
        float sc[T];
        int64_t cl[T];
#pragma unroll
        for (int t = 0; t < T; t++) {
            const int j = (wave * tcount + t) * 64 + lane, jc = j < k ? j : k - 1;
            sc[t] = scores[jc];
            cl[t] = idxs ? idxs[jc] : 0;
        }
#pragma unroll
        for (int t = 0; t < T; t++) {
            const int j = (wave * tcount + t) * 64 + lane;
            const bool in = t < tcount && j < k;
            bad |= in && (cl[t] < 0 || cl[t] >= ((int64_t)1 << kNmsClsBits));
            key[t] = in ? nms_class_key(cl[t], sc[t], j) : ~0ull;
        }
    }
    const unsigned long long mine = nms_class_key(ci, si, ic);
    bad = __syncthreads_or(bad);   // every workgroup sees every class id: all agree, the first one reports
    if (blockIdx.x == 0 && threadIdx.x == 0) *status = bad ? -1 : 0;   // -1: a class id outside the key's field, the caller takes the general route
    const int r = rank_against_slices<T, IPW, BS>(key, mine, part);
    if (threadIdx.x < IPW && i < k) {
        order[r] = i;
        cls_sorted[r] = (int64_t)(mine >> (32 + kNmsIdxBits));
        skey_sorted[r] = ((unsigned long long)desc_score_bits(si) << 32) | (unsigned)i;
        if (DIM == 4) reinterpret_cast<float4*>(boxes_sorted)[r] = make_float4(bx[0], bx[1], bx[2], bx[3]);
        else {
#pragma unroll
            for (int c = 0; c < 5; c++) boxes_sorted[(int64_t)r * 5 + c] = bx[c];
        }
    }
}
// (dets come from the SORTED boxes and the score inside the key: everything the kernel reads is indexed by the sorted position,
// nothing by a loaded value — one round trip)
__device__ __forceinline__ float score_of_desc_bits(unsigned d) {
    const unsigned u = ~d;
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
template <int T, int DIM, int IPW, int BS>
__global__ __launch_bounds__(BS) void nms_select_kernel(const float* __restrict__ boxes_sorted,
                                                               const unsigned char* __restrict__ keep_sorted,
                                                               const unsigned long long* __restrict__ skey_sorted, int k, int max_num,
                                                               int64_t* __restrict__ keep_out, float* __restrict__ dets,
                                                               int* __restrict__ status) {
    __shared__ int part[BS / 64][64];
    __shared__ int kept_waves[BS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tcount = ((k + BS - 1) / BS);
    unsigned long long key[T];
    int kept_here = 0;   // wave-uniform
    const int r = blockIdx.x * IPW + lane, rc = r < k ? r : k - 1;
    const unsigned char my_keep = keep_sorted[rc];
    const unsigned long long my_skey = skey_sorted[rc];
    float bx[5];
    load_box<DIM>(boxes_sorted, rc, bx);
    {
        unsigned char kp[T];
        unsigned long long sk[T];
#pragma unroll
        for (int t = 0; t < T; t++) {
            const int j = (wave * tcount + t) * 64 + lane, jc = j < k ? j : k - 1;
            kp[t] = keep_sorted[jc];
            sk[t] = skey_sorted[jc];
        }
#pragma unroll
        for (int t = 0; t < T; t++) {
            const int j = (wave * tcount + t) * 64 + lane;
            const bool kept = t < tcount && j < k && kp[t] != 0;
            key[t] = kept ? sk[t] : ~0ull;
            kept_here += __popcll(__builtin_amdgcn_ballot_w64(kept));
        }
    }
    if (lane == 0) kept_waves[wave] = kept_here;
    const unsigned long long mine = (r < k && my_keep != 0) ? my_skey : ~0ull;
    const int pos = rank_against_slices<T, IPW, BS>(key, mine, part);   // (its barrier also publishes kept_waves)
    if (blockIdx.x == 0 && threadIdx.x == 0 && *status >= 0) {
        int total = 0;
#pragma unroll
        for (int w = 0; w < BS / 64; w++) total += kept_waves[w];
        *status = total < max_num ? total : max_num;
    }
    if (threadIdx.x < IPW && mine != ~0ull && pos < max_num) {
        keep_out[pos] = (int64_t)(unsigned)mine;
#pragma unroll
        for (int c = 0; c < DIM; c++) dets[(int64_t)pos * (DIM + 1) + c] = bx[c];
        dets[(int64_t)pos * (DIM + 1) + DIM] = score_of_desc_bits((unsigned)(mine >> 32));
    }
}

// ---- MaxIoUAssigner epilogue: replaces overlaps.max(dim=0), overlaps.max(dim=1), the threshold steps
// and the python `for i in range(num_gts)` low-quality loop (one host sync per GT) of
// mmdet/core/bbox/assigners/max_iou_assigner.py:171-207 with three launches over the (k, n) overlaps matrix. ----
// A: one thread per column (anchor): running max / first argmax over the k rows, and per-wave row partials.
// Rows are taken 32 at a time: 32 independent coalesced loads per lane (unconditional, on clamped addresses, into a
// register array: written as `live ? ov[..] : -inf` each load sat behind its own branch and its own wait — 64 serial
// round trips per lane, 20 us instead of 9 for 64 x 98 208; requesting the NEXT round's 32 before this round's butterfly
// was measured too: 18.1 -> 17.4 us for the three kernels at 98 208 anchors, 41.2 -> 44.9 us at 392 832, not kept:
// r03r_ab_assign.log), then a transposing butterfly — at the stage
// with lane mask M a lane keeps the lower (bit clear) or upper (bit set) half of its rows and receives the partner's
// copy of that half — leaves lane L with the wave-wide maximum key of row (L >> 1) after 31 + 1 exchanges, instead of
// one 6-step wave reduction per row (192 exchanges per 32 rows).
constexpr int kAssignRows = 32;
__global__ __launch_bounds__(kBlock) void assign_cols_kernel(const float* __restrict__ ov, int k, int64_t n,
                                                            float* __restrict__ max_ov, int64_t* __restrict__ argmax_ov,
                                                            unsigned long long* __restrict__ partial, int nparts) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int part = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const bool valid = j < n;
    const int64_t jc = valid ? j : n - 1;   // clamp the address, mask the value
    const unsigned long long lane_mask = valid ? ~0ull : 0ull;   // a mask, not a select: selects here compile to 32 branches
    float best = -__builtin_inff();
    int besti = 0;
    for (int r0 = 0; r0 < k; r0 += kAssignRows) {
        unsigned long long key[kAssignRows];
        float raw[kAssignRows];
#pragma unroll
        for (int t = 0; t < kAssignRows; t++)   // unconditional (clamped addresses): all 32 loads are issued before the first wait
            raw[t] = ov[(int64_t)(r0 + t < k ? r0 + t : k - 1) * n + jc];
#pragma unroll
        for (int t = 0; t < kAssignRows; t++) {
            const int i = r0 + t;
            const float v = (valid && i < k) ? raw[t] : -__builtin_inff();
            const bool up = v > best;
            best = up ? v : best;
            besti = up ? i : besti;
            key[t] = pack_max_key(v, j) & lane_mask;   // rows past k are reduced but never written
        }
#pragma unroll
        for (int cnt = kAssignRows, m = 32; cnt > 1; cnt >>= 1, m >>= 1) {
            const bool upper = (lane & m) != 0;
            const int half = cnt >> 1;
#pragma unroll
            for (int t = 0; t < half; t++) {
                const unsigned long long mine = upper ? key[t + half] : key[t];
                const unsigned long long send = upper ? key[t] : key[t + half];
                const unsigned long long recv = __shfl_xor(send, m, 64);
                key[t] = recv > mine ? recv : mine;
            }
        }
        const unsigned long long other = __shfl_xor(key[0], 1, 64);
        const unsigned long long row_max = other > key[0] ? other : key[0];
        const int i = r0 + (lane >> 1);
        if ((lane & 1) == 0 && i < k) partial[(int64_t)i * nparts + part] = row_max;
    }
    if (valid) { max_ov[j] = best; argmax_ov[j] = besti; }
}
// B: one workgroup per row (GT): reduce the per-wave partials
__global__ __launch_bounds__(kBlock) void assign_rows_kernel(const unsigned long long* __restrict__ partial, int nparts,
                                                            float* __restrict__ gt_max, int64_t* __restrict__ gt_argmax) {
    __shared__ unsigned long long sm[kBlock / 64];
    const int i = blockIdx.x;
    unsigned long long best = 0ull;
    for (int p0 = threadIdx.x; p0 < nparts; p0 += kBlock * 8) {   // 8 independent loads per round (a repeated last partial changes no maximum)
        unsigned long long v[8];
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int p = p0 + t * kBlock;
            v[t] = partial[(int64_t)i * nparts + (p < nparts ? p : nparts - 1)];
        }
#pragma unroll
        for (int t = 0; t < 8; t++) best = v[t] > best ? v[t] : best;
    }
    best = wave_max_u64(best);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kBlock / 64; w++) best = sm[w] > best ? sm[w] : best;
        gt_max[i] = unpack_max_val(best);
        gt_argmax[i] = (int64_t)(0xffffffffu - (unsigned)best);
    }
}
// C: thresholds + low-quality matching, one thread per column; later GTs overwrite earlier ones like the python loop
__global__ __launch_bounds__(kBlock) void assign_finalize_kernel(const float* __restrict__ ov, int k, int64_t n,
                                                                const float* __restrict__ max_ov,
                                                                const int64_t* __restrict__ argmax_ov,
                                                                const float* __restrict__ gt_max,
                                                                const int64_t* __restrict__ gt_argmax, float pos_thr,
                                                                float neg_lo, float neg_hi, float min_pos,
                                                                int low_quality, int assign_all,
                                                                const int64_t* __restrict__ gt_labels,
                                                                int64_t* __restrict__ gt_inds,
                                                                int64_t* __restrict__ labels) {
    const int64_t jraw = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool valid = jraw < n;
    const int64_t j = valid ? jraw : n - 1;   // lanes past the end stay in the wave (v_readlane below reads lanes 0..31) and store nothing
    const float m = max_ov[j];
    const int64_t am = argmax_ov[j];   // unconditional: one round trip for both, not two
    int64_t a = -1;
    if (m >= neg_lo && m < neg_hi) a = 0;
    if (m >= pos_thr) a = am + 1;
    if (low_quality) {
        if (assign_all) {   // the column is re-read, 32 rows at a time: unconditional (clamped) loads, all issued before the first wait
            for (int r0 = 0; r0 < k; r0 += kAssignRows) {
                float raw[kAssignRows];
#pragma unroll
                for (int t = 0; t < kAssignRows; t++) raw[t] = ov[(int64_t)(r0 + t < k ? r0 + t : k - 1) * n + j];
                // the 32 row maxima of this round in one vector load (lane t holds row r0 + t), handed out with v_readlane;
                // a row past k or below min_pos becomes NaN, which equals nothing
                const int il = r0 + (int)(threadIdx.x & (kAssignRows - 1));
                const float gl = gt_max[il < k ? il : k - 1];
                const int gbits = __float_as_int((il < k && gl >= min_pos) ? gl : __builtin_nanf(""));
#pragma unroll
                for (int t = 0; t < kAssignRows; t++) {
                    const float g = __int_as_float(__builtin_amdgcn_readlane(gbits, t));
                    a = raw[t] == g ? r0 + t + 1 : a;
                }
            }
        } else {
            for (int i = 0; i < k; i++)
                if (gt_max[i] >= min_pos && gt_argmax[i] == j) a = i + 1;
        }
    }
    if (!valid) return;
    gt_inds[j] = a;
    if (labels) labels[j] = a > 0 ? gt_labels[a - 1] : -1;
}

// ---- fused assigner (no k x n matrix): phase 2 and 3 behind iou_pairwise_compact_kernel<.., OUT & 2> ----
// the rows' floor: what a row holds when no tile contributed (no ignore mask: IoU 0 at the shard's first column)
__device__ __forceinline__ unsigned long long row_floor(bool has_ignore, unsigned col_offset) {
    return has_ignore ? 0ull : pack_row_key(0.0f, col_offset);
}
// B' (the sharded form only): accumulators -> signed-order keys for the all-reduce; leaves the accumulators zero
__global__ __launch_bounds__(kBlock) void assign_keys_from_acc_kernel(unsigned long long* __restrict__ row_acc, int k, bool has_ignore,
                                                                     unsigned col_offset, long long* __restrict__ gt_keys) {
    const unsigned long long fl = row_floor(has_ignore, col_offset);
    for (int i = threadIdx.x; i < k; i += kBlock) {
        const unsigned long long a = row_acc[i * acc_stride(k)];
        gt_keys[i] = key_to_signed((a > fl ? a : fl) & ~1ull);
        row_acc[i * acc_stride(k)] = 0ull;
    }
}
// C': one workgroup per column tile (the tiles of phase 1).  Column maxima from the row chunks' partials, thresholds, and
// the low-quality step (max_iou_assigner.py:192-207) without the matrix: `overlaps[i, :] == gt_max[i]` can hold in this tile
//   * for gt_max[i] == 0 on every column that is not ignored (nothing overlaps GT i: every live IoU of the row is 0);
//   * for gt_max[i] == -1 on every ignored column (the whole row is ignored columns);
//   * for gt_max[i] > 0 only if the TILE's maximum of row i (row_part, still in the workspace) equals it — then at the
//     column the partial names and, only when phase 1 saw the value at a second column of the tile (bit 0 of the partial:
//     duplicated boxes, mirror-symmetric anchors), wherever a re-evaluation of the row against the tile's 256 columns with
//     the very functions phase 1 ran (same inputs, same bits) finds it.
// Later GTs overwrite earlier ones in the reference's loop: the largest matching i wins.
// FROM_ACC: the per-GT keys are phase 1's accumulators (one device); the last workgroup to finish zeroes them and the arrival
// counter for the next call.  Otherwise they are `gt_keys` (all-reduced by the caller).
template <int VARIANT, int DIM, bool ARC, bool FROM_ACC>
__global__ __launch_bounds__(kBlock) void assign_fused_finalize_kernel(const float* __restrict__ b1, int k, const float* __restrict__ b2, int n,
                                                                      int edge_arg, const unsigned long long* __restrict__ col_part, int chunks,
                                                                      const unsigned long long* __restrict__ row_part,
                                                                      const long long* __restrict__ gt_keys, unsigned long long* __restrict__ row_acc,
                                                                      bool has_ignore, unsigned col_offset,
                                                                      float pos_thr, float neg_lo, float neg_hi, float min_pos,
                                                                      int low_quality, int assign_all,
                                                                      const int64_t* __restrict__ gt_labels,
                                                                      float* __restrict__ max_ov, int64_t* __restrict__ argmax_ov,
                                                                      float* __restrict__ gt_max, int64_t* __restrict__ gt_argmax,
                                                                      int64_t* __restrict__ gt_inds, int64_t* __restrict__ labels) {
    const int edge = ARC ? (int)EDGE_ARC : edge_arg;
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x, tiles = gridDim.x;
    const int jraw = tile * kBlock + threadIdx.x;
    const bool valid = jraw < n;
    const int j = valid ? jraw : n - 1;
    const unsigned long long fl = row_floor(has_ignore, col_offset);
    auto row_key_of = [&](int i) -> unsigned long long {
        if (FROM_ACC) { const unsigned long long a = row_acc[i * acc_stride(k)]; return (a > fl ? a : fl) & ~1ull; }
        return key_from_signed(gt_keys[i]);
    };
    if (tile == 0 && gt_max) {   // the per-GT results, decoded once
        for (int i = threadIdx.x; i < k; i += kBlock) {
            const unsigned long long key = row_key_of(i);
            gt_max[i] = unpack_max_val(key);
            if (gt_argmax) gt_argmax[i] = (int64_t)row_key_index(key);
        }
    }
    unsigned long long ck = 0ull;
    for (int c0 = 0; c0 < chunks; c0 += 8) {   // 8 independent loads per round (a repeated last chunk changes no maximum)
        unsigned long long v[8];
#pragma unroll
        for (int t = 0; t < 8; t++) v[t] = col_part[(int64_t)(c0 + t < chunks ? c0 + t : chunks - 1) * n + j];
#pragma unroll
        for (int t = 0; t < 8; t++) ck = v[t] > ck ? v[t] : ck;
    }
    const float m = unpack_max_val(ck);
    const int64_t am = (int64_t)(0xffffffffu - (unsigned)ck);
    const bool ign = m < 0.0f;   // only an ignored column has a negative maximum
    int64_t a = -1;
    if (m >= neg_lo && m < neg_hi) a = 0;
    if (m >= pos_thr) a = am + 1;
    if (low_quality) {
        int best = -1;
        if (!assign_all) {
            for (int i = 0; i < k; i++) {
                const unsigned long long key = row_key_of(i);
                if (unpack_max_val(key) >= min_pos && row_key_index(key) == col_offset + (unsigned)j) best = i;
            }
        } else {
            float a5[5] = {0.0f, 0.0f, 1.0f, 1.0f, 0.0f};
            bool have_box = false;
            CullBox ca{};
            ColatTrig ct{};
            for (int r0 = 0; r0 < k; r0 += 64) {
                const int il = r0 + lane, ic = il < k ? il : k - 1;
                const unsigned long long gk = row_key_of(ic);
                const float g = unpack_max_val(gk);
                const bool on = il < k && g >= min_pos;
                const unsigned long long zero = __builtin_amdgcn_ballot_w64(on && g == 0.0f);
                const unsigned long long neg = __builtin_amdgcn_ballot_w64(on && g == -1.0f);
                const unsigned long long pk = row_part[(int64_t)ic * tiles + tile];
                const bool here = on && g > 0.0f && (unsigned)(pk >> 32) == (unsigned)(gk >> 32);   // the row's maximum lives in this tile
                unsigned long long rec = __builtin_amdgcn_ballot_w64(here && (pk & 1ull));              // ... maybe at several columns
                unsigned long long one = __builtin_amdgcn_ballot_w64(here && !(pk & 1ull));             // ... at the one the partial names
                if (!ign && zero) best = r0 + 63 - __builtin_clzll(zero);
                if (ign && neg) { const int t = r0 + 63 - __builtin_clzll(neg); best = t > best ? t : best; }
                while (one) {   // wave-uniform; descending, so the first hit is the largest row
                    const int t = 63 - __builtin_clzll(one);
                    one &= ~(1ull << t);
                    const unsigned col = row_key_index(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)pk, t)));
                    if (valid && col == col_offset + (unsigned)j && r0 + t > best) best = r0 + t;
                }
                while (rec) {   // wave-uniform: rows whose maximum lives in this tile
                    const int t = __builtin_ctzll(rec);
                    rec &= rec - 1;
                    if (!have_box) {
                        if (valid) load_box<DIM>(b2, j, a5);
                        ca = cull_box(a5, edge);
                        ct = colat_trig(a5[1], 2);
                        have_box = true;
                    }
                    float g5[5];
                    load_box<DIM>(b1, r0 + t, g5);
                    const CullBox cg = cull_box(g5, edge);
                    float v = 0.0f;
                    if (!cull_pair(cg, ca)) v = lean_finish<VARIANT, DIM, 1>(g5, a5, MODE_IOU, edge, colat_trig(g5[1], 1), ct);
                    const float gm = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g), t));
                    if (!ign && v == gm && r0 + t > best) best = r0 + t;
                }
            }
        }
        if (best >= 0) a = best + 1;
    }
    if (valid) {
        max_ov[j] = m;
        if (argmax_ov) argmax_ov[j] = am;
        gt_inds[j] = a;
        if (labels) labels[j] = a > 0 ? gt_labels[a - 1] : -1;
    }
    if (FROM_ACC) {   // every read of the accumulators above has returned (its value was used); the last arrival cleans up.
        // Two levels of arrival counters (a returning atomic per workgroup on ONE address serialises): the last of each group of
        // 32 tiles reports to the top counter, the last group to report zeroes the accumulators and every counter.
        __shared__ unsigned ticket;
        unsigned long long* counters = row_acc + acc_words(k);   // [0]: top, [1 + g]: group g, kAccLine words apart
        const int groups = (int)ticket_groups(tiles), grp = tile / kTicketGroup;
        const int members = grp == groups - 1 ? tiles - grp * kTicketGroup : kTicketGroup;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned t = __hip_atomic_fetch_add((unsigned*)(counters + (int64_t)(1 + grp) * kAccLine), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned last = 0;
            if ((int)t == members - 1)
                last = (int)__hip_atomic_fetch_add((unsigned*)counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1;
            ticket = last;
        }
        __syncthreads();
        if (ticket) {
            for (int i = threadIdx.x; i < k; i += kBlock) row_acc[i * acc_stride(k)] = 0ull;
            for (int i = threadIdx.x; i <= groups; i += kBlock) counters[(int64_t)i * kAccLine] = 0ull;
        }
    }
}

// ---- adjoint of the Sph2Pob transform: gradients of the planar boxes -> gradients of the spherical boxes ----
template <int VARIANT, int DIM>
__global__ __launch_bounds__(kBlock) void transform_bwd_kernel(const float* __restrict__ b1, const float* __restrict__ b2,
                                                              const float* __restrict__ g1, const float* __restrict__ g2,
                                                              float* __restrict__ gb1, float* __restrict__ gb2, int64_t n,
                                                              int edge, int jitter) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float x[5], y[5], gx[5], gy[5], p[5], q[5];
    load_box<DIM>(b1, i, x);
    load_box<DIM>(b2, i, y);
#pragma unroll
    for (int k = 0; k < 5; k++) { p[k] = g1[i * 5 + k]; q[k] = g2[i * 5 + k]; }
    pair_transform_bwd<VARIANT, DIM>(x, y, p, q, edge, jitter != 0, gx, gy);
#pragma unroll
    for (int k = 0; k < DIM; k++) { gb1[i * DIM + k] = gx[k]; gb2[i * DIM + k] = gy[k]; }
}

// adjoint of the reference-order transforms by forward-mode differentiation (legacy, rbb_angle='project')
template <int VARIANT, int DIM>
__global__ __launch_bounds__(kBlock) void transform_bwd_dual_kernel(const float* __restrict__ b1, const float* __restrict__ b2,
                                                                   const float* __restrict__ g1, const float* __restrict__ g2,
                                                                   float* __restrict__ gb1, float* __restrict__ gb2, int64_t n,
                                                                   int edge, int angle, int jitter) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float x[5], y[5], gx[5], gy[5], p[5], q[5];
    load_box<DIM>(b1, i, x);
    load_box<DIM>(b2, i, y);
#pragma unroll
    for (int k = 0; k < 5; k++) { p[k] = g1[i * 5 + k]; q[k] = g2[i * 5 + k]; }
    transform_bwd_dual<VARIANT, DIM>(x, y, p, q, edge, angle, jitter != 0, gx, gy);
#pragma unroll
    for (int k = 0; k < DIM; k++) { gb1[i * DIM + k] = gx[k]; gb2[i * DIM + k] = gy[k]; }
}

int check_common(int box_dim, int variant_flags, int edge, int angle) {
    const int variant = variant_flags & 0xff;
    if (variant_flags & ~(0xff | SPH2POB_FLAG_REFERENCE_ORDER | SPH2POB_FLAG_ROBUST_PARALLEL | SPH2POB_FLAG_NAIVE_TAN)) return SPH2POB_ERR_OPTION;
    if ((variant_flags & SPH2POB_FLAG_NAIVE_TAN) && variant != SPH2POB_VARIANT_NAIVE) return SPH2POB_ERR_OPTION;
    if (box_dim != 4 && box_dim != 5) return SPH2POB_ERR_DIM;
    if (variant < 0 || variant > SPH2POB_VARIANT_NAIVE || edge < 0 || edge > 2 || angle < 0 || angle > 1) return SPH2POB_ERR_OPTION;
    if (variant >= SPH2POB_VARIANT_LEGACY && variant <= SPH2POB_VARIANT_FOV_IOU && box_dim == 5)
        return SPH2POB_ERR_DIM;  // BFoV-only variants
    return SPH2POB_OK;
}

int launch_status() {
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

struct AlignedLaunch {
    const float *b1, *b2; float* out; int64_t n; int mode, edge, angle; hipStream_t s; bool fast = true;
    template <int V, int D> int run() {
        dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
        if (V <= 2 && n < ((int64_t)1 << 31) - 1024 && !g_no_compact) {
            // persistent-style grid.  Measured on MI355X (`tools/ab.py run --workload aligned ... label:SPH2POB_WGS_PER_CU=k`; round 1: tools/sweep_slices.sh): every CU must hold the same number of
            // workgroups (1 303 workgroups = 5.09 per CU take 12 % longer than 1 536 = 6 per CU); 6 per CU (24 waves per CU)
            // is the best or within noise of the best from 125 k to 8 M pairs; small launches want one slice per wave
            // rather than full survivor stacks.  Hence: whole multiples of the CU count, at most 6 per CU (and never
            // more than the LDS admits), at least one 64-pair slice per wave.
            const int64_t kCUs = cu_count();
            int64_t resident = (160 * 1024) / queue_lds_bytes(D);
            if (resident > (D == 4 ? 6 : 5)) resident = D == 4 ? 6 : 5;   // registers (__launch_bounds__) would admit 7 / 5
            int64_t slices = (n + 63) / 64;
            int64_t wgs = (slices + 3) / 4;
            if (g_slices_per_wave > 0) wgs = (slices + 4 * g_slices_per_wave - 1) / (4 * g_slices_per_wave);
            else if (wgs > kCUs) { wgs = (wgs + kCUs - 1) / kCUs * kCUs; if (wgs > kCUs * resident) wgs = kCUs * resident; }
            if (g_wgs_per_cu > 0) wgs = kCUs * g_wgs_per_cu;
            if (wgs < 1) wgs = 1;
            constexpr int VV = V > 2 ? 0 : V;
            const bool ref_finish = !fast || V == 2 || angle != SPH2POB_ANGLE_EQUATOR;
            const int edge_k = edge | (angle << 8);
#define SPH_PIPE(PF, ARC, REF) hipLaunchKernelGGL((iou_aligned_compact_kernel<VV, D, PF, ARC, REF>), dim3((unsigned)wgs), dim3(kBlock), 0, s, b1, b2, out, (int)n, mode, edge_k)
            if (!ref_finish && V < 2 && !g_persistent) {   // the default: one-round chunk kernel
                const unsigned cw = (unsigned)((n + kBlock * kChunkSlices - 1) / (kBlock * kChunkSlices));
                // cull phase at a higher wave priority while (nearly) the whole grid is resident at once: see the kernel
                const int edge_k = (edge | (angle << 8)) | (D == 4 && (int64_t)cw <= kCUs * 8 * 2 && !g_no_prio ? 0x10000 : 0);
                if (edge == SPH2POB_EDGE_ARC) hipLaunchKernelGGL((iou_aligned_chunk_kernel<VV, D, true, kChunkSlices>), dim3(cw), dim3(kBlock), 0, s, b1, b2, out, (int)n, mode, edge_k);
                else hipLaunchKernelGGL((iou_aligned_chunk_kernel<VV, D, false, kChunkSlices>), dim3(cw), dim3(kBlock), 0, s, b1, b2, out, (int)n, mode, edge_k);
            } else
            if (ref_finish) { if (wgs > kCUs * 4) wgs = kCUs * 4; SPH_PIPE(true, false, true); }   // reference-order finish: 4 waves per SIMD
            else if (edge == SPH2POB_EDGE_ARC) { if (g_prefetch) SPH_PIPE(true, true, false); else SPH_PIPE(false, true, false); }
            else { if (g_prefetch) SPH_PIPE(true, false, false); else SPH_PIPE(false, false, false); }
#undef SPH_PIPE
        } else if (fast && V < 2 && angle == SPH2POB_ANGLE_EQUATOR)
            hipLaunchKernelGGL((iou_aligned_kernel<V >= 2 ? 0 : V, D, true>), grid, dim3(kBlock), 0, s, b1, b2, out, n, mode, edge, angle);
        else if (V >= 5 && fast)  // unbiased / naive: `fast` selects the default (double) arithmetic of the unbiased IoU
            hipLaunchKernelGGL((iou_aligned_kernel<V >= 5 ? V : 5, D, true>), grid, dim3(kBlock), 0, s, b1, b2, out, n, mode, edge, angle);
        else
            hipLaunchKernelGGL((iou_aligned_kernel<V, D, false>), grid, dim3(kBlock), 0, s, b1, b2, out, n, mode, edge, angle);
        return launch_status();
    }
};
// rows per workgroup of iou_pairwise_compact_kernel: enough to amortise the per-column setup and fill the survivor stacks,
// few enough that the grid holds thousands of workgroups; with the tail-first dispatch order, 64 GT
// (profiles/r03y_ab_pairwise_rows.log): 98 208 anchors 18.5 us at 8 rows, 19.9 at 12, 20.2 at 16, 33.9 at 32; 392 832
// anchors 49.1 us at 8, 47.0 at 12, 42.6 at 16, 41.1 at 22, 42.0 at 32 => about 4 096 workgroups, at least 8 rows, chunks of
// equal size
static int64_t pairwise_rows_per_wg(int64_t m, int64_t n) {
    const int64_t col_tiles = (n + kBlock - 1) / kBlock;
    int64_t rpw = g_pw_rows > 0 ? g_pw_rows : (m * col_tiles) / 4096;
    if (rpw < 8 && g_pw_rows <= 0) rpw = 8;
    if (rpw < 4) rpw = 4;
    if (rpw > kPwRows) rpw = kPwRows;
    if (rpw > m) rpw = m;
    if (g_pw_rows <= 0) rpw = (m + (m + rpw - 1) / rpw - 1) / ((m + rpw - 1) / rpw);   // 64 rows: 23 -> 3 chunks of 22 / 22 / 20
    return rpw;
}
struct PairwiseLaunch {
    const float* b1; int64_t m; const float* b2; int64_t n; float* out; int mode, edge, angle; hipStream_t s; bool fast = true;
    template <int V, int D> int run() {
        if (fast && V < 2 && angle == SPH2POB_ANGLE_EQUATOR && n < ((int64_t)1 << 31) - kBlock && m <= (int64_t)65535 * 4 &&
            !g_no_compact) {
            const int64_t col_tiles = (n + kBlock - 1) / kBlock;
            const int64_t rpw = pairwise_rows_per_wg(m, n);
            dim3 grid((unsigned)col_tiles, (unsigned)((m + rpw - 1) / rpw));
            if (edge == SPH2POB_EDGE_ARC)
                hipLaunchKernelGGL((iou_pairwise_compact_kernel<V >= 2 ? 0 : V, D, true, 1>), grid, dim3(kBlock), 0, s, b1, (int)m, b2, (int)n,
                                   out, mode, edge, (int)rpw, (const unsigned char*)nullptr, (unsigned long long*)nullptr,
                                   (unsigned long long*)nullptr, 0u, (unsigned long long*)nullptr);
            else
                hipLaunchKernelGGL((iou_pairwise_compact_kernel<V >= 2 ? 0 : V, D, false, 1>), grid, dim3(kBlock), 0, s, b1, (int)m, b2, (int)n,
                                   out, mode, edge, (int)rpw, (const unsigned char*)nullptr, (unsigned long long*)nullptr,
                                   (unsigned long long*)nullptr, 0u, (unsigned long long*)nullptr);
            return launch_status();
        }
        // grid.y is limited to 65535 rows per launch: walk the rows in slabs
        const int64_t kMaxRows = 65535;
        for (int64_t r0 = 0; r0 < m; r0 += kMaxRows) {
            int64_t rows = m - r0 < kMaxRows ? m - r0 : kMaxRows;
            dim3 grid((unsigned)((n + kBlock - 1) / kBlock), (unsigned)rows);
            if (fast && V < 2 && angle == SPH2POB_ANGLE_EQUATOR)
                hipLaunchKernelGGL((iou_pairwise_kernel<V >= 2 ? 0 : V, D, true>), grid, dim3(kBlock), 0, s, b1 + r0 * D, rows,
                                   b2, n, out + r0 * n, mode, edge, angle);
            else if (V >= 5 && fast)
                hipLaunchKernelGGL((iou_pairwise_kernel<V >= 5 ? V : 5, D, true>), grid, dim3(kBlock), 0, s, b1 + r0 * D, rows,
                                   b2, n, out + r0 * n, mode, edge, angle);
            else
                hipLaunchKernelGGL((iou_pairwise_kernel<V, D, false>), grid, dim3(kBlock), 0, s, b1 + r0 * D, rows, b2, n,
                                   out + r0 * n, mode, edge, angle);
            int rc = launch_status();
            if (rc) return rc;
        }
        return SPH2POB_OK;
    }
};
// the fused assigner's two halves (closed-form standard / efficient only: the kernels that carry the reductions)
struct AssignWs { unsigned long long *row_acc, *col_part, *row_part; int64_t chunks, tiles; };
static AssignWs assign_ws(void* workspace, void* state, int64_t k, int64_t n) {
    AssignWs w;
    w.tiles = (n + kBlock - 1) / kBlock;
    const int64_t rpw = pairwise_rows_per_wg(k, n);
    w.chunks = (k + rpw - 1) / rpw;
    w.row_acc = (unsigned long long*)state;   // k accumulators + the arrival counter: zero between calls
    w.col_part = (unsigned long long*)workspace;
    w.row_part = w.col_part + w.chunks * n;
    return w;
}
struct AssignReduceLaunch {
    const float* b1; int64_t m; const float* b2; int64_t n; float* out; int edge; const unsigned char* ignore; unsigned col_offset;
    long long* gt_keys /* NULL: leave the keys in the accumulators */; void* workspace; void* state; hipStream_t s; bool fast = true;
    template <int V, int D> int run() {
        if constexpr (V >= 2) return SPH2POB_ERR_OPTION;
        else {
            if (!fast) return SPH2POB_ERR_OPTION;
            const AssignWs w = assign_ws(workspace, state, m, n);
            const int64_t rpw = pairwise_rows_per_wg(m, n);
            dim3 grid((unsigned)w.tiles, (unsigned)w.chunks);
#define SPH_AR(ARC, OUT) hipLaunchKernelGGL((iou_pairwise_compact_kernel<V, D, ARC, OUT>), grid, dim3(kBlock), 0, s, b1, (int)m, b2, (int)n, out, \
                                           (int)MODE_IOU, edge, (int)rpw, ignore, w.col_part, w.row_part, col_offset, w.row_acc)
            if (edge == SPH2POB_EDGE_ARC) { if (out) SPH_AR(true, 3); else SPH_AR(true, 2); }
            else { if (out) SPH_AR(false, 3); else SPH_AR(false, 2); }
#undef SPH_AR
            if (gt_keys)
                hipLaunchKernelGGL(assign_keys_from_acc_kernel, dim3(1), dim3(kBlock), 0, s, w.row_acc, (int)m, ignore != nullptr, col_offset, gt_keys);
            return launch_status();
        }
    }
};
struct AssignFinalizeLaunch {
    const float* b1; int64_t m; const float* b2; int64_t n; int edge; unsigned col_offset; const long long* gt_keys /* NULL: the accumulators */;
    bool has_ignore;
    float pos, neg_lo, neg_hi, min_pos; int low_quality, assign_all; const int64_t* gt_labels; float* max_ov; int64_t* argmax_ov;
    float* gt_max; int64_t* gt_argmax; int64_t* gt_inds; int64_t* labels; void* workspace; void* state; hipStream_t s; bool fast = true;
    template <int V, int D> int run() {
        if constexpr (V >= 2) return SPH2POB_ERR_OPTION;
        else {
            if (!fast) return SPH2POB_ERR_OPTION;
            const AssignWs w = assign_ws(workspace, state, m, n);
#define SPH_AF(ARC, ACC) hipLaunchKernelGGL((assign_fused_finalize_kernel<V, D, ARC, ACC>), dim3((unsigned)w.tiles), dim3(kBlock), 0, s, b1, (int)m, b2, \
                                           (int)n, edge, w.col_part, (int)w.chunks, w.row_part, gt_keys, w.row_acc, has_ignore, col_offset, pos, neg_lo,  \
                                           neg_hi, min_pos, low_quality, assign_all, gt_labels, max_ov, argmax_ov, gt_max, gt_argmax, gt_inds, labels)
            if (edge == SPH2POB_EDGE_ARC) { if (gt_keys) SPH_AF(true, false); else SPH_AF(true, true); }
            else { if (gt_keys) SPH_AF(false, false); else SPH_AF(false, true); }
#undef SPH_AF
            return launch_status();
        }
    }
};
struct TransformLaunch {
    const float *b1, *b2; float *o1, *o2; int64_t n; int edge, angle, jitter; hipStream_t s; bool fast = true;
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
    if (mode < 0 || mode > 1 || ((variant & 0xff) >= SPH2POB_VARIANT_UNBIASED && mode != SPH2POB_MODE_IOU)) return SPH2POB_ERR_OPTION;
    if (n < 0 || n > kMaxElems) return SPH2POB_ERR_SIZE;
    if (n == 0) return SPH2POB_OK;
    if (!b1 || !b2 || !out) return SPH2POB_ERR_NULL;
    if (variant & SPH2POB_FLAG_NAIVE_TAN) edge = SPH2POB_EDGE_TANGENT;
    return dispatch(variant, box_dim, AlignedLaunch{b1, b2, out, n, mode, edge, angle, (hipStream_t)stream});
}

int sph2pob_iou_pairwise_f32(const float* b1, int64_t m, const float* b2, int64_t n, float* out, int box_dim,
                             int variant, int mode, int edge, int angle, void* stream) {
    int rc = check_common(box_dim, variant, edge, angle);
    if (rc) return rc;
    if (mode < 0 || mode > 1 || ((variant & 0xff) >= SPH2POB_VARIANT_UNBIASED && mode != SPH2POB_MODE_IOU)) return SPH2POB_ERR_OPTION;
    if (m < 0 || n < 0 || n > kMaxElems || m > kMaxElems) return SPH2POB_ERR_SIZE;
    if (m == 0 || n == 0) return SPH2POB_OK;
    if (!b1 || !b2 || !out) return SPH2POB_ERR_NULL;
    if (variant & SPH2POB_FLAG_NAIVE_TAN) edge = SPH2POB_EDGE_TANGENT;
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

int sph2pob_planar_iou_f32(const float* p1, int64_t m, const float* p2, int64_t n, float* out, int aligned, int mode,
                           void* stream) {
    if (mode < 0 || mode > 1) return SPH2POB_ERR_OPTION;
    if (m < 0 || n < 0 || m > kMaxElems || n > kMaxElems || (aligned && m != n)) return SPH2POB_ERR_SIZE;
    if (m == 0 || n == 0) return SPH2POB_OK;
    if (!p1 || !p2 || !out) return SPH2POB_ERR_NULL;
    hipStream_t s = (hipStream_t)stream;
    const unsigned bx = (unsigned)((n + kBlock - 1) / kBlock);
    if (aligned) {
        hipLaunchKernelGGL(planar_iou_kernel, dim3(bx), dim3(kBlock), 0, s, p1, m, p2, n, out, 1, mode);
        return launch_status();
    }
    const int64_t kMaxRows = 65535;   // grid.y limit: walk the rows in slabs
    for (int64_t r0 = 0; r0 < m; r0 += kMaxRows) {
        const int64_t rows = m - r0 < kMaxRows ? m - r0 : kMaxRows;
        hipLaunchKernelGGL(planar_iou_kernel, dim3(bx, (unsigned)rows), dim3(kBlock), 0, s, p1 + r0 * 5, rows, p2, n,
                           out + r0 * n, 0, mode);
        const int rc = launch_status();
        if (rc) return rc;
    }
    return SPH2POB_OK;
}

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

int sph2pob_nms_max_boxes(void) { return kNmsMaxWords * 64 - 64; }  // per class segment (unaligned: L/64 + 2 words)

static int64_t nms_row_words(int64_t k, int64_t max_segment) {
    int64_t full = (k + 63) / 64, seg = (max_segment >> 6) + 2;  // an unaligned segment of L boxes spans <= L/64 + 2 words
    return seg < full ? seg : full;
}
int64_t sph2pob_nms_workspace_bytes(int64_t k) { return k * nms_row_words(k, k) * 8; }
int64_t sph2pob_nms_segmented_workspace_bytes(int64_t k, int64_t max_segment) {
    return k * nms_row_words(k, max_segment < 1 ? 1 : max_segment) * 8;
}

static int nms_check_options(int box_dim, int variant_flags) {
    const int variant = variant_flags & 0xff;
    // SPH2POB_FLAG_ROBUST_PARALLEL is accepted and has no effect here (a near-parallel pair is far above any threshold)
    if (variant_flags & ~(0xff | SPH2POB_FLAG_REFERENCE_ORDER | SPH2POB_FLAG_ROBUST_PARALLEL | SPH2POB_FLAG_NAIVE_TAN)) return SPH2POB_ERR_OPTION;
    if ((variant_flags & SPH2POB_FLAG_NAIVE_TAN) && variant != SPH2POB_VARIANT_NAIVE) return SPH2POB_ERR_OPTION;
    if (box_dim != 4 && box_dim != 5) return SPH2POB_ERR_DIM;
    if (variant != SPH2POB_VARIANT_STANDARD && variant != SPH2POB_VARIANT_EFFICIENT && variant != SPH2POB_VARIANT_UNBIASED &&
        variant != SPH2POB_VARIANT_NAIVE)
        return SPH2POB_ERR_OPTION;
    return SPH2POB_OK;
}
// mask + sweep on boxes sorted by (class, -score): the two launches every NMS entry point shares
static int nms_mask_and_sweep(const float* boxes_sorted, const int64_t* cls_sorted, int64_t k, int box_dim, int variant_flags,
                              float iou_threshold, int words, unsigned long long* mask, unsigned char* keep, hipStream_t s) {
    const int variant = variant_flags & 0xff;
    const bool fast = !(variant_flags & SPH2POB_FLAG_REFERENCE_ORDER);
    const int wpb = kBlock / 64;
    dim3 grid((unsigned)((k + wpb - 1) / wpb));
#define SPH_NMS_LAUNCH(V, D, F) \
    hipLaunchKernelGGL((nms_mask_kernel<V, D, F>), grid, dim3(kBlock), 0, s, boxes_sorted, cls_sorted, k, words, iou_threshold, mask, \
                       (variant_flags & SPH2POB_FLAG_NAIVE_TAN) ? (int)EDGE_TANGENT : (int)EDGE_ARC)
#define SPH_NMS_COMPACT(V, D) \
    hipLaunchKernelGGL((nms_mask_compact_kernel<V, D>), grid, dim3(kBlock), 0, s, boxes_sorted, cls_sorted, k, words, iou_threshold, mask)
    const bool compact = fast && !g_no_compact && k < ((int64_t)1 << 31) - 64 &&
                         (variant == SPH2POB_VARIANT_EFFICIENT || variant == SPH2POB_VARIANT_STANDARD);
    if (compact) {
        if (variant == SPH2POB_VARIANT_EFFICIENT) { if (box_dim == 4) SPH_NMS_COMPACT(1, 4); else SPH_NMS_COMPACT(1, 5); }
        else { if (box_dim == 4) SPH_NMS_COMPACT(0, 4); else SPH_NMS_COMPACT(0, 5); }
    } else if (variant == SPH2POB_VARIANT_EFFICIENT) {
        if (box_dim == 4) { if (fast) SPH_NMS_LAUNCH(1, 4, true); else SPH_NMS_LAUNCH(1, 4, false); }
        else { if (fast) SPH_NMS_LAUNCH(1, 5, true); else SPH_NMS_LAUNCH(1, 5, false); }
    } else if (variant == SPH2POB_VARIANT_UNBIASED) {  // sph_nms.py:11-12
        if (box_dim == 4) { if (fast) SPH_NMS_LAUNCH(5, 4, true); else SPH_NMS_LAUNCH(5, 4, false); }
        else { if (fast) SPH_NMS_LAUNCH(5, 5, true); else SPH_NMS_LAUNCH(5, 5, false); }
    } else if (variant == SPH2POB_VARIANT_NAIVE) {     // sph_nms.py:13-14
        if (box_dim == 4) SPH_NMS_LAUNCH(6, 4, false); else SPH_NMS_LAUNCH(6, 5, false);
    } else {
        if (box_dim == 4) { if (fast) SPH_NMS_LAUNCH(0, 4, true); else SPH_NMS_LAUNCH(0, 4, false); }
        else { if (fast) SPH_NMS_LAUNCH(0, 5, true); else SPH_NMS_LAUNCH(0, 5, false); }
    }
#undef SPH_NMS_LAUNCH
#undef SPH_NMS_COMPACT
    int rc = launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(nms_sweep_kernel, dim3((unsigned)((k + kSweepCands - 1) / kSweepCands)), dim3(kSweepBlock), 0, s, mask, cls_sorted, k,
                       words, keep);
    return launch_status();
}

int sph2pob_nms_segmented_f32(const float* boxes_sorted, const int64_t* cls_sorted, int64_t k, int box_dim,
                              int variant_flags, float iou_threshold, int64_t max_segment, void* workspace,
                              unsigned char* keep, void* stream) {
    int rc = nms_check_options(box_dim, variant_flags);
    if (rc) return rc;
    if (k < 0 || k > ((int64_t)1 << 31) - 64 || max_segment < 0 || max_segment > sph2pob_nms_max_boxes())
        return SPH2POB_ERR_SIZE;
    if (k == 0) return SPH2POB_OK;
    if (!boxes_sorted || !workspace || !keep) return SPH2POB_ERR_NULL;
    if (!cls_sorted && max_segment < k) return SPH2POB_ERR_SIZE;  // one segment: it is k long
    return nms_mask_and_sweep(boxes_sorted, cls_sorted, k, box_dim, variant_flags, iou_threshold,
                              (int)nms_row_words(k, max_segment < 1 ? 1 : max_segment), (unsigned long long*)workspace, keep,
                              (hipStream_t)stream);
}

// workspace of the host-free form: sorted boxes | classes | permutation | keep flags | mask matrix (all 256-byte aligned)
struct BatchedNmsWs { float* boxes; int64_t* cls; int* order; unsigned long long* skey; unsigned char* keep; unsigned long long* mask; int64_t bytes; int words; };
static BatchedNmsWs batched_nms_ws(void* workspace, int64_t k, int box_dim) {
    auto up = [](int64_t x) { return (x + 255) / 256 * 256; };
    BatchedNmsWs w;
    char* p = (char*)workspace;
    w.words = (int)nms_row_words(k, k);
    int64_t off = 0;
    w.boxes = (float*)(p + off); off += up(k * box_dim * 4);
    w.cls = (int64_t*)(p + off); off += up(k * 8);
    w.order = (int*)(p + off); off += up(k * 4);
    w.skey = (unsigned long long*)(p + off); off += up(k * 8);
    w.keep = (unsigned char*)(p + off); off += up(k);
    w.mask = (unsigned long long*)(p + off); off += up(k * (int64_t)w.words * 8);
    w.bytes = off;
    return w;
}
int sph2pob_batched_nms_max_boxes(void) { return 1 << kNmsIdxBits; }
int64_t sph2pob_batched_nms_workspace_bytes(int64_t k, int box_dim) {
    return k > 0 && k <= sph2pob_batched_nms_max_boxes() ? batched_nms_ws(nullptr, k, box_dim).bytes : 0;
}
int sph2pob_batched_nms_f32(const float* boxes, const float* scores, const int64_t* idxs, int64_t k, int box_dim, int variant_flags,
                            float iou_threshold, int64_t max_num, void* workspace, int64_t* keep, float* dets, int* status,
                            void* stream) {
    int rc = nms_check_options(box_dim, variant_flags);
    if (rc) return rc;
    if (k < 0 || k > sph2pob_batched_nms_max_boxes() || max_num < 0) return SPH2POB_ERR_SIZE;
    if (!status) return SPH2POB_ERR_NULL;
    hipStream_t s = (hipStream_t)stream;
    if (k == 0 || max_num == 0) return hipMemsetAsync(status, 0, sizeof(int), s) == hipSuccess ? SPH2POB_OK : (int)hipGetLastError();
    if (!boxes || !scores || !workspace || !keep || !dets) return SPH2POB_ERR_NULL;
    const BatchedNmsWs w = batched_nms_ws(workspace, k, box_dim);
    const int kk = (int)k, mx = (int)(max_num < k ? max_num : k);
    // T keys per lane x BS threads cover the candidates, IPW boxes per workgroup, chosen so that the grid is at most 256
    // workgroups: with ~150 VGPRs per lane a CU holds ONE of these workgroups at a time, and 313 of them (16 boxes each at
    // K = 5 000) ran in two rounds — 12 us per launch where 6 144 candidates or fewer now take one round
#define SPH_PREP(T, D, I, B) hipLaunchKernelGGL((nms_prepare_kernel<T, D, I, B>), dim3((unsigned)((k + I - 1) / I)), dim3(B), 0, s, boxes, scores, idxs, kk, \
                                                w.boxes, w.cls, w.order, w.skey, status)
#define SPH_SEL(T, D, I, B) hipLaunchKernelGGL((nms_select_kernel<T, D, I, B>), dim3((unsigned)((k + I - 1) / I)), dim3(B), 0, s, (const float*)w.boxes, \
                                               (const unsigned char*)w.keep, (const unsigned long long*)w.skey, kk, mx, keep, dets, status)
#define SPH_BY_SIZE(M, D) do { if (k <= 2048) M(4, D, 16, 512); else if (k <= 4096) M(8, D, 16, 512); else if (k <= 6144) M(12, D, 24, 512); \
                               else if (k <= 8192) M(16, D, 32, 512); else if (k <= 12288) M(12, D, 48, 1024); else M(16, D, 64, 1024); } while (0)
    if (box_dim == 4) SPH_BY_SIZE(SPH_PREP, 4); else SPH_BY_SIZE(SPH_PREP, 5);
    rc = launch_status();
    if (rc) return rc;
    rc = nms_mask_and_sweep(w.boxes, idxs ? w.cls : nullptr, k, box_dim, variant_flags, iou_threshold, w.words, w.mask, w.keep, s);
    if (rc) return rc;
    if (box_dim == 4) SPH_BY_SIZE(SPH_SEL, 4); else SPH_BY_SIZE(SPH_SEL, 5);
#undef SPH_PREP
#undef SPH_SEL
#undef SPH_BY_SIZE
    return launch_status();
}

int sph2pob_nms_f32(const float* boxes_sorted, const int64_t* cls_sorted, int64_t k, int box_dim, int variant_flags,
                    float iou_threshold, void* workspace, unsigned char* keep, void* stream) {
    if (k > sph2pob_nms_max_boxes()) return SPH2POB_ERR_SIZE;
    return sph2pob_nms_segmented_f32(boxes_sorted, cls_sorted, k, box_dim, variant_flags, iou_threshold, k, workspace, keep,
                                     stream);
}

int64_t sph2pob_assign_workspace_bytes(int64_t k, int64_t n) {
    int64_t nparts = ((n + kBlock - 1) / kBlock) * (kBlock / 64);
    return k * nparts * 8;
}

int sph2pob_assign_f32(const float* overlaps, int64_t k, int64_t n, float pos_iou_thr, float neg_iou_lo,
                       float neg_iou_hi, float min_pos_iou, int match_low_quality, int gt_max_assign_all,
                       const int64_t* gt_labels, float* max_overlaps, int64_t* argmax_overlaps, float* gt_max_overlaps,
                       int64_t* gt_argmax_overlaps, int64_t* assigned_gt_inds, int64_t* assigned_labels, void* workspace,
                       void* stream) {
    if (k <= 0 || n <= 0 || k > 0x7fffffff || n > kMaxElems || n > (int64_t)0xfffffffe) return SPH2POB_ERR_SIZE;
    if (!overlaps || !max_overlaps || !argmax_overlaps || !gt_max_overlaps || !gt_argmax_overlaps || !assigned_gt_inds ||
        !workspace || (assigned_labels && !gt_labels))
        return SPH2POB_ERR_NULL;
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((n + kBlock - 1) / kBlock);
    const int nparts = (int)(blocks * (kBlock / 64));
    unsigned long long* partial = (unsigned long long*)workspace;
    hipLaunchKernelGGL(assign_cols_kernel, dim3(blocks), dim3(kBlock), 0, s, overlaps, (int)k, n, max_overlaps,
                       argmax_overlaps, partial, nparts);
    hipLaunchKernelGGL(assign_rows_kernel, dim3((unsigned)k), dim3(kBlock), 0, s, partial, nparts, gt_max_overlaps,
                       gt_argmax_overlaps);
    hipLaunchKernelGGL(assign_finalize_kernel, dim3(blocks), dim3(kBlock), 0, s, overlaps, (int)k, n, max_overlaps,
                       argmax_overlaps, gt_max_overlaps, gt_argmax_overlaps, pos_iou_thr, neg_iou_lo, neg_iou_hi,
                       min_pos_iou, match_low_quality, gt_max_assign_all, gt_labels, assigned_gt_inds, assigned_labels);
    return launch_status();
}

int64_t sph2pob_iou_assign_workspace_bytes(int64_t k, int64_t n) {
    if (k <= 0 || n <= 0) return 0;
    const AssignWs w = assign_ws(nullptr, nullptr, k, n);
    return (w.chunks * n + k * w.tiles) * 8;
}
int64_t sph2pob_iou_assign_state_bytes(int64_t k, int64_t n) {
    return k > 0 && n > 0 ? (acc_words(k) + (1 + ticket_groups((n + kBlock - 1) / kBlock)) * kAccLine) * 8 : 0;
}

static int assign_fused_check(int64_t k, int64_t n, int box_dim, int variant, int edge, int64_t col_offset) {
    int rc = check_common(box_dim, variant, edge, 0);
    if (rc) return rc;
    if ((variant & 0xff) > SPH2POB_VARIANT_EFFICIENT || (variant & SPH2POB_FLAG_REFERENCE_ORDER)) return SPH2POB_ERR_OPTION;
    if (k <= 0 || n <= 0 || n >= ((int64_t)1 << 31) - kBlock || k > (int64_t)65535 * 4 || col_offset < 0 ||
        col_offset + n > (int64_t)0x7ffffffe)
        return SPH2POB_ERR_SIZE;
    return SPH2POB_OK;
}

int sph2pob_iou_assign_reduce_f32(const float* gt, int64_t k, const float* boxes, int64_t n, int box_dim, int variant, int edge,
                                  const unsigned char* ignore, int64_t col_offset, float* overlaps, int64_t* gt_keys,
                                  void* workspace, void* state, void* stream) {
    int rc = assign_fused_check(k, n, box_dim, variant, edge, col_offset);
    if (rc) return rc;
    if (!gt || !boxes || !gt_keys || !workspace || !state) return SPH2POB_ERR_NULL;
    return dispatch(variant, box_dim, AssignReduceLaunch{gt, k, boxes, n, overlaps, edge, ignore, (unsigned)col_offset,
                                                         (long long*)gt_keys, workspace, state, (hipStream_t)stream});
}

int sph2pob_iou_assign_finalize_f32(const float* gt, int64_t k, const float* boxes, int64_t n, int box_dim, int variant, int edge,
                                    int64_t col_offset, const int64_t* gt_keys, float pos_iou_thr, float neg_iou_lo,
                                    float neg_iou_hi, float min_pos_iou, int match_low_quality, int gt_max_assign_all,
                                    const int64_t* gt_labels, float* max_overlaps, int64_t* argmax_overlaps,
                                    float* gt_max_overlaps, int64_t* gt_argmax_overlaps, int64_t* assigned_gt_inds,
                                    int64_t* assigned_labels, void* workspace, void* stream) {
    int rc = assign_fused_check(k, n, box_dim, variant, edge, col_offset);
    if (rc) return rc;
    if (!gt || !boxes || !gt_keys || !workspace || !max_overlaps || !assigned_gt_inds || (assigned_labels && !gt_labels))
        return SPH2POB_ERR_NULL;
    return dispatch(variant, box_dim,
                    AssignFinalizeLaunch{gt, k, boxes, n, edge, (unsigned)col_offset, (const long long*)gt_keys, false, pos_iou_thr,
                                         neg_iou_lo, neg_iou_hi, min_pos_iou, match_low_quality, gt_max_assign_all, gt_labels,
                                         max_overlaps, argmax_overlaps, gt_max_overlaps, gt_argmax_overlaps, assigned_gt_inds,
                                         assigned_labels, workspace, nullptr, (hipStream_t)stream});
}

int sph2pob_iou_assign_f32(const float* gt, int64_t k, const float* boxes, int64_t n, int box_dim, int variant, int edge,
                           const unsigned char* ignore, float* overlaps, float pos_iou_thr, float neg_iou_lo, float neg_iou_hi,
                           float min_pos_iou, int match_low_quality, int gt_max_assign_all, const int64_t* gt_labels,
                           float* max_overlaps, int64_t* argmax_overlaps, float* gt_max_overlaps, int64_t* gt_argmax_overlaps,
                           int64_t* assigned_gt_inds, int64_t* assigned_labels, void* workspace, void* state, void* stream) {
    int rc = assign_fused_check(k, n, box_dim, variant, edge, 0);
    if (rc) return rc;
    if (!gt || !boxes || !workspace || !state || !max_overlaps || !assigned_gt_inds || (assigned_labels && !gt_labels)) return SPH2POB_ERR_NULL;
    // two launches: the per-GT keys stay in the workspace's accumulators, the finalize pass reads and clears them
    rc = dispatch(variant, box_dim, AssignReduceLaunch{gt, k, boxes, n, overlaps, edge, ignore, 0u, nullptr, workspace, state, (hipStream_t)stream});
    if (rc) return rc;
    return dispatch(variant, box_dim,
                    AssignFinalizeLaunch{gt, k, boxes, n, edge, 0u, nullptr, ignore != nullptr, pos_iou_thr, neg_iou_lo, neg_iou_hi,
                                         min_pos_iou, match_low_quality, gt_max_assign_all, gt_labels, max_overlaps, argmax_overlaps,
                                         gt_max_overlaps, gt_argmax_overlaps, assigned_gt_inds, assigned_labels, workspace, state,
                                         (hipStream_t)stream});
}

int sph2pob_transform_bwd_f32(const float* b1, const float* b2, const float* grad_planar1, const float* grad_planar2,
                              float* grad_b1, float* grad_b2, int64_t n, int box_dim, int variant, int edge, int jitter,
                              void* stream) {
    int rc = check_common(box_dim, variant, edge, 0);
    if (rc) return rc;
    if ((variant & 0xff) == SPH2POB_VARIANT_LEGACY) return SPH2POB_ERR_OPTION;
    if (n < 0 || n > kMaxElems) return SPH2POB_ERR_SIZE;
    if (n == 0) return SPH2POB_OK;
    if (!b1 || !b2 || !grad_planar1 || !grad_planar2 || !grad_b1 || !grad_b2) return SPH2POB_ERR_NULL;
    dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    hipStream_t s = (hipStream_t)stream;
#define SPH_TBWD(V, D) \
    hipLaunchKernelGGL((transform_bwd_kernel<V, D>), grid, dim3(kBlock), 0, s, b1, b2, grad_planar1, grad_planar2, grad_b1, grad_b2, n, edge, jitter)
    if ((variant & 0xff) == SPH2POB_VARIANT_STANDARD) { if (box_dim == 4) SPH_TBWD(0, 4); else SPH_TBWD(0, 5); }
    else { if (box_dim == 4) SPH_TBWD(1, 4); else SPH_TBWD(1, 5); }
#undef SPH_TBWD
    return launch_status();
}

int sph2pob_transform_bwd_general_f32(const float* b1, const float* b2, const float* grad_planar1,
                                      const float* grad_planar2, float* grad_b1, float* grad_b2, int64_t n, int box_dim,
                                      int variant, int edge, int angle, int jitter, void* stream) {
    int rc = check_common(box_dim, variant, edge, angle);
    if (rc) return rc;
    const int v = variant & 0xff;
    if (v > SPH2POB_VARIANT_LEGACY) return SPH2POB_ERR_OPTION;
    if (n < 0 || n > kMaxElems) return SPH2POB_ERR_SIZE;
    if (n == 0) return SPH2POB_OK;
    if (!b1 || !b2 || !grad_planar1 || !grad_planar2 || !grad_b1 || !grad_b2) return SPH2POB_ERR_NULL;
    dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
    hipStream_t s = (hipStream_t)stream;
#define SPH_TBWDD(V, D) \
    hipLaunchKernelGGL((transform_bwd_dual_kernel<V, D>), grid, dim3(kBlock), 0, s, b1, b2, grad_planar1, grad_planar2, grad_b1, grad_b2, n, edge, angle, jitter)
    if (v == SPH2POB_VARIANT_LEGACY) SPH_TBWDD(2, 4);
    else if (v == SPH2POB_VARIANT_STANDARD) { if (box_dim == 4) SPH_TBWDD(0, 4); else SPH_TBWDD(0, 5); }
    else { if (box_dim == 4) SPH_TBWDD(1, 4); else SPH_TBWDD(1, 5); }
#undef SPH_TBWDD
    return launch_status();
}

#if defined(SPH_STAMPS)
int sph2pob_debug_set_stamps(void* buffer) {
    unsigned long long* p = (unsigned long long*)buffer;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p));
}
#endif

}  // extern "C"

// libsph2pob_hip.so — aligned IoU (the dominant kernels), the Sph2Pob transform and its adjoint, planar IoU: kernels + C-ABI
// launchers (include/sph2pob_hip.h).  gfx950 only.

#include "sph2pob_kernels_common.hpp"

namespace {


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
struct TransformLaunch {
    const float *b1, *b2; float *o1, *o2; int64_t n; int edge, angle, jitter; hipStream_t s; bool fast = true;
    template <int V, int D> int run() {
        dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
        hipLaunchKernelGGL((transform_kernel<V, D>), grid, dim3(kBlock), 0, s, b1, b2, o1, o2, n, edge, angle, jitter);
        return launch_status();
    }
};

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

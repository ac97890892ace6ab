// libsph2pob_hip.so — pairwise IoU (the assigner call pattern) and the MaxIoUAssigner epilogues, matrix and fused: kernels +
// C-ABI launchers (include/sph2pob_hip.h).  gfx950 only.

#include "sph2pob_kernels_common.hpp"

namespace {


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
            const bool up = v > best || (v != v && best == best);   // torch.max: the first NaN is the maximum and stays
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

}  // namespace

extern "C" {


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

}  // extern "C"

// libsph2pob_hip.so — greedy per-class NMS on Sph2Pob IoU (sorted-input form and the host-free batched form): kernels + C-ABI
// launchers (include/sph2pob_hip.h).  gfx950 only.

#include "sph2pob_kernels_common.hpp"

namespace {


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
            hit = !(pair_iou_sel<VARIANT, DIM, FAST>(x, y, MODE_IOU, edge, ANGLE_EQUATOR) <= thr);   // the reference keeps `iou <= thr` (sph_nms.py:72): a NaN IoU suppresses
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
        // VARIANT_UNBIASED (the spherical rectangles' exact intersection area, fp64): the rectangles lie inside the caps the cull
        // compares — a rectangle's corners are at atan(sqrt(tan^2(a/2) + tan^2(b/2))) <= sqrt(a^2 + b^2) / 2 from its centre —
        // so a culled pair has intersection 0.  For RBFoV that is IoU 0; the BFoV form returns (0 + 1e-8) / (A1 + A2 - 1e-8)
        // (unbiased_iou_bfov.py:200), which stays at or below thr as long as A1 + A2 >= 1e-8 (1 + 1 / thr): tested on a lower
        // bound of the areas (A >= 4 a b / pi^2), twice over; degenerate boxes and thr <= 0 are never culled.
        const float xab = fminf(fmaxf(x[2], 0.0f), 180.0f) * fminf(fmaxf(x[3], 0.0f), 180.0f);   // (out-of-range / NaN extents count as 0)
        const float ub_need = thr > 0.0f ? 2e-8f * (1.0f + 1.0f / thr) / 1.2345679e-4f : __builtin_inff();   // in deg^2: 4 / 180^2
        (void)xab; (void)ub_need;
        int* st = stack[wave];
        int count = 0;
        auto finish_one = [&](int j) {
            float y[5];
            load_box<DIM>(boxes, j, y);
            float v;
            if constexpr (VARIANT == VARIANT_UNBIASED) v = unbiased_pair_iou<DIM, false>(x, y);
            else v = lean_finish<VARIANT, DIM, 2>(x, y, MODE_IOU, EDGE_ARC, xt);
            if (!(v <= thr)) {   // `iou <= thr` keeps: NaN suppresses
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
                if constexpr (VARIANT == VARIANT_UNBIASED && DIM == 4)
                    surv |= !(fmaf(fminf(fmaxf(y[2], 0.0f), 180.0f), fminf(fmaxf(y[3], 0.0f), 180.0f), xab) >= ub_need);
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
// (Blocks of 128 or 256 rows — two / four diagonal words per row, half / a quarter of the barriers — were built and measured in
// round 3, bit-equal: 145 / 162 us against 147 for one class of 5 000, 39.7 / 45.0 against 38.2 us for 37 classes
// (profiles/r06t_sweep_rb.log).  Not kept.  Nor were, later in the round: the same with a 1 024-thread workgroup so that the OR stage's
// tasks fit one round (144 us, r07f_sweep.log), and dense copies of each row's diagonal word and the word after it so that the
// serial chain's prefetch is 8 cache lines per block instead of 128 (148 us, r07g_diag.log).  What IS known: one class of 1 000 /
// 2 500 / 5 000 / 10 000 / 20 000 boxes sweeps in 16.5 / 40.3 / 79.2 / 184 / 523 us — ~1 us per 64 rows up to 5 000 boxes
// whatever the block size, the workgroup's size or the layout of the words the chain reads; VALU issue 0.6 % (r07b PMC).)
// (And with the OR stage two blocks deep ON TOP of the prefetch fix below — wave 0 carrying two words per row, the workers' loads
// issued in one iteration and consumed in the next across the LDS-only barrier, static register sets: bit-equal, 146.1 -> 143.5 us
// for one class of 5 000, 38.1 -> 42.1 us for 37 classes (r07i_sweep2.log).  Not kept: what a block costs is wave 0's own
// instruction chain — a lone wave issues a dependent instruction every ~9 cycles, 14 per kept row, ~7 kept rows per block, plus
// the LDS read, the ballot, the store and the barrier.)
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
        // Three register sets in rotation (the block loop is unrolled by three so that the sets are named statically): block b's
        // words are requested right AFTER block b - 2 is resolved and used two barriers later.  (Round 2 requested block b + 1's
        // words at the top of iteration b: the s_waitcnt vmcnt(0) in front of block b's first v_readlane then waited for them as
        // well — loads retire in order — and every block paid a full trip to the mask, ~1 us, whatever else was done to it.
        // The barrier is the LDS-only one for the same reason: __syncthreads() waits for every outstanding load and store.)
        unsigned long long dg[3], af[3];
        dg[0] = load_word(0, 0); af[0] = load_word(0, 1);
        dg[1] = load_word(1, 1); af[1] = load_word(1, 2);
        dg[2] = 0ull; af[2] = 0ull;
        unsigned long long carry = 0ull;   // wave 0, uniform: what block b - 1's kept rows remove in word b
        for (int b0 = 0; b0 <= b_last; b0 += 3) {
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const int b = b0 + u;
                if (b > b_last) break;   // workgroup-uniform
                if (wave == 0) {
                    const int64_t row = (base + b) * 64 + lane;
                    const bool mine = row >= s && row < seg_end;
                    const unsigned dlo = (unsigned)dg[u], dhi = (unsigned)(dg[u] >> 32);
                    const unsigned alo = (unsigned)af[u], ahi = (unsigned)(af[u] >> 32);
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
                    dg[(u + 2) % 3] = load_word(b + 2, b + 2);   // this set held block b - 1: consumed
                    af[(u + 2) % 3] = load_word(b + 2, b + 3);
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
                        for (int q = 0; q < 16; q++) v[q] = ((bits >> q) & 1u) ? col[(int64_t)q * words] : 0ull;
                        unsigned long long acc = 0ull;
#pragma unroll
                        for (int q = 0; q < 16; q++) acc |= v[q];
                        if (acc) atomicOr(&removed[w], acc);
                    }
                }
                // LDS-only barrier: the waves talk through removed[] and kept_sh[] alone (the workers have consumed their loads
                // before their ds_or; nobody reads keep[]), and wave 0's prefetch must stay in flight across it
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
        __syncthreads();
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
    // the order of torch's device sort (cub's radix keys): by bit pattern — +NaN first, -NaN last — except that -0 is +0
    // (equal scores keep the index order)
    unsigned u = __float_as_uint(v + 0.0f);
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


}  // namespace

extern "C" {


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
                         (variant == SPH2POB_VARIANT_EFFICIENT || variant == SPH2POB_VARIANT_STANDARD || variant == SPH2POB_VARIANT_UNBIASED);
    if (compact) {
        if (variant == SPH2POB_VARIANT_UNBIASED) { if (box_dim == 4) SPH_NMS_COMPACT(5, 4); else SPH_NMS_COMPACT(5, 5); }
        else if (variant == SPH2POB_VARIANT_EFFICIENT) { if (box_dim == 4) SPH_NMS_COMPACT(1, 4); else SPH_NMS_COMPACT(1, 5); }
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

}  // extern "C"

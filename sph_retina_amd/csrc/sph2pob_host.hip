// libsph2pob_host.so — the CPU twins of the C ABI ("CPU twins with the same names suffixed _cpu").
//
// The reference's operators run on CPU tensors as well (tests/test_all_ious.py:88-104 runs every IoU with device='cpu';
// sphdet/iou/sph_iou_calculator.py:107-108 forces unbiased_iou to the CPU; MaxIoUAssigner's gpu_assign_thr moves the
// assignment to the CPU, mmdet/core/bbox/assigners/max_iou_assigner.py:100-110): this library serves those calls from the
// PRODUCT'S OWN arithmetic — the very __host__ __device__ functions of sph2pob_{device,fast,loss,unbiased}.hpp that the
// kernels run, instantiated for the host — on a small thread pool.  It is not the oracle and shares no code with it
// (oracle/ is the checker; tests/test_capi_symbols.py guards the separation).  Results follow the host's libm where the
// device uses ocml: within the fp32 noise documented in DESIGN.md §3, not bit-identical to the kernels.
//
// Every entry point has the signature of its HIP twin (the trailing stream argument is ignored: the call is synchronous) and
// returns the same error codes.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "../../include/sph2pob_hip.h"
#include "sph2pob_device.hpp"
#include "sph2pob_loss.hpp"
#include "sph2pob_fast.hpp"
#include "sph2pob_unbiased.hpp"
#include "sph2pob_coder.hpp"

namespace {

using namespace sph2pob;

int cpu_threads() {
    static int n = 0;
    if (n == 0) {
        const char* e = getenv("SPH2POB_CPU_THREADS");
        int v = e ? atoi(e) : 0;
        if (v <= 0) {
            v = (int)std::thread::hardware_concurrency();
            cpu_set_t set;
            if (sched_getaffinity(0, sizeof(set), &set) == 0) v = std::min(v, CPU_COUNT(&set));
        }
        n = std::max(1, std::min(v, 256));
    }
    return n;
}

// f(lo, hi) over [0, n) in contiguous chunks, one per thread; small jobs stay on the caller's thread
template <class F>
void parallel_for(int64_t n, int64_t grain, F&& f) {
    int t = (int)std::min<int64_t>(cpu_threads(), (n + grain - 1) / std::max<int64_t>(grain, 1));
    if (t <= 1) { if (n > 0) f((int64_t)0, n); return; }
    std::vector<std::thread> pool;
    pool.reserve(t - 1);
    const int64_t per = (n + t - 1) / t;
    for (int k = 1; k < t; k++) {
        const int64_t lo = k * per, hi = std::min(n, lo + per);
        if (lo < hi) pool.emplace_back([&f, lo, hi] { f(lo, hi); });
    }
    f((int64_t)0, std::min(n, per));
    for (auto& th : pool) th.join();
}

template <int DIM>
inline void load_box(const float* p, int64_t i, float (&b)[5]) {
    for (int k = 0; k < 5; k++) b[k] = k < DIM ? p[i * DIM + k] : 0.0f;
}

// the selection the kernels make (sph2pob_kernels_common.hpp: pair_iou_sel; sph2pob_iou.hip: the launch rules of AlignedLaunch)
template <int V, int DIM>
inline float pair_iou_any(const float (&x)[5], const float (&y)[5], bool fast, int mode, int edge, int angle) {
    if constexpr (V == VARIANT_UNBIASED) return fast ? unbiased_pair_iou<DIM, false>(x, y) : unbiased_pair_iou<DIM, true>(x, y);
    else if constexpr (V == VARIANT_NAIVE) return naive_iou<DIM>(x, y, edge == EDGE_TANGENT);
    else if constexpr (V < 2) return (fast && angle == ANGLE_EQUATOR) ? pair_iou_fast<V, DIM>(x, y, mode, edge) : pair_iou<V, DIM>(x, y, mode, edge, angle);
    else return pair_iou<V, DIM>(x, y, mode, edge, angle);
}

template <class F>
int dispatch(int variant_flags, int box_dim, F&& f) {
    const int v = variant_flags & 0xff;
    f.fast = !(variant_flags & SPH2POB_FLAG_REFERENCE_ORDER);
    switch (v) {
        case SPH2POB_VARIANT_STANDARD: return box_dim == 4 ? f.template run<0, 4>() : f.template run<0, 5>();
        case SPH2POB_VARIANT_EFFICIENT: return box_dim == 4 ? f.template run<1, 4>() : f.template run<1, 5>();
        case SPH2POB_VARIANT_SPH_IOU: return f.template run<3, 4>();
        case SPH2POB_VARIANT_FOV_IOU: return f.template run<4, 4>();
        case SPH2POB_VARIANT_UNBIASED: return box_dim == 4 ? f.template run<5, 4>() : f.template run<5, 5>();
        case SPH2POB_VARIANT_NAIVE: return box_dim == 4 ? f.template run<6, 4>() : f.template run<6, 5>();
        default: return f.template run<2, 4>();
    }
}

int check_common(int box_dim, int variant_flags, int edge, int angle) {
    const int variant = variant_flags & 0xff;
    if (variant_flags & ~(0xff | SPH2POB_FLAG_REFERENCE_ORDER | SPH2POB_FLAG_ROBUST_PARALLEL | SPH2POB_FLAG_NAIVE_TAN)) return SPH2POB_ERR_OPTION;
    if ((variant_flags & SPH2POB_FLAG_NAIVE_TAN) && variant != SPH2POB_VARIANT_NAIVE) return SPH2POB_ERR_OPTION;
    if (box_dim != 4 && box_dim != 5) return SPH2POB_ERR_DIM;
    if (variant < 0 || variant > SPH2POB_VARIANT_NAIVE || edge < 0 || edge > 2 || angle < 0 || angle > 1) return SPH2POB_ERR_OPTION;
    if (variant >= SPH2POB_VARIANT_LEGACY && variant <= SPH2POB_VARIANT_FOV_IOU && box_dim == 5) return SPH2POB_ERR_DIM;
    return SPH2POB_OK;
}
constexpr int64_t kMaxElems = (int64_t)1 << 38;

struct Aligned {
    const float *b1, *b2; float* out; int64_t n; int mode, edge, angle; bool fast = true;
    template <int V, int D> int run() {
        parallel_for(n, 2048, [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; i++) {
                float x[5], y[5];
                load_box<D>(b1, i, x);
                load_box<D>(b2, i, y);
                out[i] = pair_iou_any<V, D>(x, y, fast, mode, edge, angle);
            }
        });
        return SPH2POB_OK;
    }
};
struct Pairwise {
    const float* b1; int64_t m; const float* b2; int64_t n; float* out; int mode, edge, angle; bool fast = true;
    template <int V, int D> int run() {
        parallel_for(m * n, 2048, [&](int64_t lo, int64_t hi) {
            for (int64_t e = lo; e < hi; e++) {
                float x[5], y[5];
                load_box<D>(b1, e / n, x);
                load_box<D>(b2, e % n, y);
                out[e] = pair_iou_any<V, D>(x, y, fast, mode, edge, angle);
            }
        });
        return SPH2POB_OK;
    }
};
struct Transform {
    const float *b1, *b2; float *o1, *o2; int64_t n; int edge, angle, jitter; bool fast = true;
    template <int V, int D> int run() {
        if constexpr (V > 2) return SPH2POB_ERR_OPTION;
        else {
            parallel_for(n, 2048, [&](int64_t lo, int64_t hi) {
                for (int64_t i = lo; i < hi; i++) {
                    float x[5], y[5];
                    load_box<D>(b1, i, x);
                    load_box<D>(b2, i, y);
                    if (jitter) jitter_spherical<D>(x, y);
                    PBox p1, p2;
                    transform<V, D>(x, y, edge, angle, p1, p2);
                    if (jitter) jitter_rotated(p1, p2);
                    float* q1 = o1 + i * 5;
                    float* q2 = o2 + i * 5;
                    q1[0] = p1.x; q1[1] = p1.y; q1[2] = p1.w; q1[3] = p1.h; q1[4] = p1.a;
                    q2[0] = p2.x; q2[1] = p2.y; q2[2] = p2.w; q2[3] = p2.h; q2[4] = p2.a;
                }
            });
            return SPH2POB_OK;
        }
    }
};

// ---- loss ----
template <int DIM>
inline float element_weight(const float* w, int wd, int64_t i) {   // sph2pob_loss.hip: element_weight
    if (!w) return 1.0f;
    if (wd == 1) return w[i];
    float s = 0.0f;
    for (int k = 0; k < DIM; k++) s += w[i * DIM + k];
    if (DIM == 4) return (s + s / 4.0f) / 5.0f;
    return s / (float)DIM;
}
int loss_check(const float* weight, int weight_dim, int64_t n, int box_dim, int loss_mode_flags) {
    const int loss_mode = loss_mode_flags & 0xff;
    if (loss_mode_flags & ~(0xff | SPH2POB_FLAG_REFERENCE_ORDER)) return SPH2POB_ERR_OPTION;
    if (box_dim != 4 && box_dim != 5) return SPH2POB_ERR_DIM;
    if (loss_mode < 0 || loss_mode > 3) return SPH2POB_ERR_OPTION;
    if (weight && weight_dim != 1 && weight_dim != box_dim) return SPH2POB_ERR_OPTION;
    if (n < 0 || n > kMaxElems) return SPH2POB_ERR_SIZE;
    return SPH2POB_OK;
}
// one pass over the pairs: loss element (times w), IoU, gradients (times g) as asked for; returns the sum of the elements
template <int DIM, bool FAST>
double loss_pass(const float* pred, const float* target, const float* weight, int wd, float scale, const float* grad_out, int grad_stride,
                 float* loss, float* iou, float* gpred, float* gtarget, int64_t n, int mode, float eps) {
    const int t = cpu_threads();
    std::vector<double> partial((size_t)t + 1, 0.0);
    const int64_t per = (n + t - 1) / std::max(t, 1);
    parallel_for(n, 1024, [&](int64_t lo, int64_t hi) {
        double acc = 0.0;
        for (int64_t i = lo; i < hi; i++) {
            float x[5], y[5], gx[5], gy[5], io = 0.0f;
            const float w = scale * element_weight<DIM>(weight, wd, i);
            const float g = grad_out ? grad_out[i * grad_stride] * w : w;
            load_box<DIM>(pred, i, x);
            load_box<DIM>(target, i, y);
            const bool need_grad = gpred || gtarget;
            const float l = need_grad ? pair_loss<DIM, true, FAST>(x, y, mode, eps, &io, gx, gy)
                                      : pair_loss<DIM, false, FAST>(x, y, mode, eps, &io, gx, gy);
            const float lw = w == 0.0f ? 0.0f : l * w;   // the kernels skip all-zero-weight waves: a zero weight gives exact zeros
            if (loss) loss[i] = lw;
            if (iou) iou[i] = io;
            acc += lw;
            for (int k = 0; k < DIM; k++) {
                if (gpred) gpred[i * DIM + k] = g == 0.0f ? 0.0f : g * gx[k];
                if (gtarget) gtarget[i * DIM + k] = g == 0.0f ? 0.0f : g * gy[k];
            }
        }
        partial[(size_t)std::min<int64_t>(lo / std::max<int64_t>(per, 1), t)] += acc;
    });
    double s = 0.0;
    for (double v : partial) s += v;   // fixed order: reproducible for a given thread count
    return s;
}
template <class... A>
double loss_pass_sel(int box_dim, bool fast, A... a) {
    if (box_dim == 4) return fast ? loss_pass<4, true>(a...) : loss_pass<4, false>(a...);
    return fast ? loss_pass<5, true>(a...) : loss_pass<5, false>(a...);
}

// greedy NMS per class segment (sph2pob_nms_segmented_f32)
struct NmsRun {
    const float* boxes; const int64_t* cls; int64_t k; float thr; unsigned char* keep; int edge; bool fast = true;
    template <int V, int D> int run() {
        if constexpr (V == 2 || V == 3 || V == 4) return SPH2POB_ERR_OPTION;
        else {
            // class segments are independent: one thread sweeps a segment (the greedy dependency is serial inside it)
            std::vector<int64_t> starts;
            for (int64_t i = 0; i < k; i++)
                if (i == 0 || (cls && cls[i] != cls[i - 1])) starts.push_back(i);
            starts.push_back(k);
            const int64_t segs = (int64_t)starts.size() - 1;
            parallel_for(segs, 1, [&](int64_t lo, int64_t hi) {
                for (int64_t s = lo; s < hi; s++) {
                    const int64_t a = starts[(size_t)s], b = starts[(size_t)s + 1];
                    std::vector<unsigned char> removed((size_t)(b - a), 0);
                    for (int64_t i = a; i < b; i++) {
                        keep[i] = !removed[(size_t)(i - a)];
                        if (!keep[i]) continue;
                        float x[5];
                        load_box<D>(boxes, i, x);
                        for (int64_t j = i + 1; j < b; j++) {
                            if (removed[(size_t)(j - a)]) continue;
                            float y[5];
                            load_box<D>(boxes, j, y);
                            if (!(pair_iou_any<V, D>(x, y, fast, MODE_IOU, edge, ANGLE_EQUATOR) <= thr)) removed[(size_t)(j - a)] = 1;
                        }
                    }
                }
            });
            return SPH2POB_OK;
        }
    }
};

}  // namespace

extern "C" {

int sph2pob_host_abi_version(void) { return 1; }
int sph2pob_host_threads(void) { return cpu_threads(); }

int sph2pob_iou_aligned_f32_cpu(const float* b1, const float* b2, float* out, int64_t n, int box_dim, int variant, int mode, int edge,
                                int angle, void*) {
    int rc = check_common(box_dim, variant, edge, angle);
    if (rc) return rc;
    if (mode < 0 || mode > 1 || ((variant & 0xff) >= SPH2POB_VARIANT_UNBIASED && mode != SPH2POB_MODE_IOU)) return SPH2POB_ERR_OPTION;
    if (n < 0 || n > kMaxElems) return SPH2POB_ERR_SIZE;
    if (n == 0) return SPH2POB_OK;
    if (!b1 || !b2 || !out) return SPH2POB_ERR_NULL;
    if (variant & SPH2POB_FLAG_NAIVE_TAN) edge = SPH2POB_EDGE_TANGENT;
    return dispatch(variant, box_dim, Aligned{b1, b2, out, n, mode, edge, angle});
}

int sph2pob_iou_pairwise_f32_cpu(const float* b1, int64_t m, const float* b2, int64_t n, float* out, int box_dim, int variant, int mode,
                                 int edge, int angle, void*) {
    int rc = check_common(box_dim, variant, edge, angle);
    if (rc) return rc;
    if (mode < 0 || mode > 1 || ((variant & 0xff) >= SPH2POB_VARIANT_UNBIASED && mode != SPH2POB_MODE_IOU)) return SPH2POB_ERR_OPTION;
    if (m < 0 || n < 0 || n > kMaxElems || m > kMaxElems) return SPH2POB_ERR_SIZE;
    if (m == 0 || n == 0) return SPH2POB_OK;
    if (!b1 || !b2 || !out) return SPH2POB_ERR_NULL;
    if (variant & SPH2POB_FLAG_NAIVE_TAN) edge = SPH2POB_EDGE_TANGENT;
    return dispatch(variant, box_dim, Pairwise{b1, m, b2, n, out, mode, edge, angle});
}

int sph2pob_transform_f32_cpu(const float* b1, const float* b2, float* planar1, float* planar2, int64_t n, int box_dim, int variant,
                              int edge, int angle, int jitter, void*) {
    int rc = check_common(box_dim, variant, edge, angle);
    if (rc) return rc;
    if (n < 0 || n > kMaxElems) return SPH2POB_ERR_SIZE;
    if (n == 0) return SPH2POB_OK;
    if (!b1 || !b2 || !planar1 || !planar2) return SPH2POB_ERR_NULL;
    return dispatch(variant, box_dim, Transform{b1, b2, planar1, planar2, n, edge, angle, jitter});
}

int sph2pob_planar_iou_f32_cpu(const float* p1, int64_t m, const float* p2, int64_t n, float* out, int aligned, int mode, void*) {
    if (mode < 0 || mode > 1) return SPH2POB_ERR_OPTION;
    if (m < 0 || n < 0 || m > kMaxElems || n > kMaxElems || (aligned && m != n)) return SPH2POB_ERR_SIZE;
    if (m == 0 || n == 0) return SPH2POB_OK;
    if (!p1 || !p2 || !out) return SPH2POB_ERR_NULL;
    const int64_t total = aligned ? n : m * n;
    parallel_for(total, 4096, [&](int64_t lo, int64_t hi) {
        for (int64_t e = lo; e < hi; e++) {
            const float* a = p1 + (aligned ? e : e / n) * 5;
            const float* b = p2 + (aligned ? e : e % n) * 5;
            out[e] = planar_iou(PBox{a[0], a[1], a[2], a[3], a[4]}, PBox{b[0], b[1], b[2], b[3], b[4]}, mode);
        }
    });
    return SPH2POB_OK;
}

int sph2pob_loss_fwd_f32_cpu(const float* pred, const float* target, const float* weight, int weight_dim, float scale, float* loss,
                             float* iou, int64_t n, int box_dim, int loss_mode_flags, float eps, void*) {
    int rc = loss_check(weight, weight_dim, n, box_dim, loss_mode_flags);
    if (rc) return rc;
    if (n == 0) return SPH2POB_OK;
    if (!pred || !target || !loss) return SPH2POB_ERR_NULL;
    loss_pass_sel(box_dim, !(loss_mode_flags & SPH2POB_FLAG_REFERENCE_ORDER), pred, target, weight, weight_dim, scale, (const float*)nullptr, 0,
                  loss, iou, (float*)nullptr, (float*)nullptr, n, loss_mode_flags & 0xff, eps);
    return SPH2POB_OK;
}

int sph2pob_loss_bwd_f32_cpu(const float* pred, const float* target, const float* weight, int weight_dim, const float* grad_out,
                             int grad_stride, float scale, float* grad_pred, float* grad_target, int64_t n, int box_dim,
                             int loss_mode_flags, float eps, void*) {
    int rc = loss_check(weight, weight_dim, n, box_dim, loss_mode_flags);
    if (rc) return rc;
    if (grad_stride != 0 && grad_stride != 1) return SPH2POB_ERR_OPTION;
    if (n == 0) return SPH2POB_OK;
    if (!pred || !target || !grad_out || !grad_pred) return SPH2POB_ERR_NULL;
    loss_pass_sel(box_dim, !(loss_mode_flags & SPH2POB_FLAG_REFERENCE_ORDER), pred, target, weight, weight_dim, scale, grad_out, grad_stride,
                  (float*)nullptr, (float*)nullptr, grad_pred, grad_target, n, loss_mode_flags & 0xff, eps);
    return SPH2POB_OK;
}

int sph2pob_loss_fwd_sum_f32_cpu(const float* pred, const float* target, const float* weight, int weight_dim, float scale, float* out,
                                 float* workspace, int64_t n, int box_dim, int loss_mode_flags, float eps, void*) {
    (void)workspace;
    int rc = loss_check(weight, weight_dim, n, box_dim, loss_mode_flags);
    if (rc) return rc;
    if (!out || (n > 0 && (!pred || !target))) return SPH2POB_ERR_NULL;
    // (the elements are summed unscaled and the scale applied once, as the HIP form does)
    const double s = n ? loss_pass_sel(box_dim, !(loss_mode_flags & SPH2POB_FLAG_REFERENCE_ORDER), pred, target, weight, weight_dim, 1.0f,
                                       (const float*)nullptr, 0, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, n,
                                       loss_mode_flags & 0xff, eps)
                       : 0.0;
    out[0] = (float)(s * (double)scale);
    return SPH2POB_OK;
}

int sph2pob_loss_fwd_grad_f32_cpu(const float* pred, const float* target, const float* weight, int weight_dim, float scale, float* loss,
                                  float* out_sum, float* workspace, float* grad_pred, float* grad_target, int64_t n, int box_dim,
                                  int loss_mode_flags, float eps, void*) {
    (void)workspace;
    int rc = loss_check(weight, weight_dim, n, box_dim, loss_mode_flags);
    if (rc) return rc;
    if (n > 0 && (!pred || !target || !grad_pred)) return SPH2POB_ERR_NULL;
    const double s = n ? loss_pass_sel(box_dim, !(loss_mode_flags & SPH2POB_FLAG_REFERENCE_ORDER), pred, target, weight, weight_dim, scale,
                                       (const float*)nullptr, 0, loss, (float*)nullptr, grad_pred, grad_target, n, loss_mode_flags & 0xff, eps)
                       : 0.0;
    if (out_sum) out_sum[0] = (float)s;
    return SPH2POB_OK;
}

int sph2pob_loss_grad_scale_f32_cpu(const float* stash, const float* grad_out, int grad_stride, float* out, int64_t n, int box_dim, void*) {
    if (n < 0 || (box_dim != 4 && box_dim != 5) || (grad_stride != 0 && grad_stride != 1)) return SPH2POB_ERR_OPTION;
    if (n == 0) return SPH2POB_OK;
    if (!stash || !grad_out || !out) return SPH2POB_ERR_NULL;
    if (grad_stride == 0 && out == stash && grad_out[0] == 1.0f) return SPH2POB_OK;
    const int64_t total = n * box_dim;
    for (int64_t e = 0; e < total; e++) out[e] = stash[e] * grad_out[grad_stride ? e / box_dim : 0];
    return SPH2POB_OK;
}

int sph2pob_sum_f32_cpu(const float* x, int64_t n, float scale, float* out, float* workspace, void*) {
    (void)workspace;
    if (n < 0) return SPH2POB_ERR_SIZE;
    if (!out || (n > 0 && !x)) return SPH2POB_ERR_NULL;
    double s = 0.0;
    for (int64_t i = 0; i < n; i++) s += x[i];
    out[0] = (float)(s * (double)scale);
    return SPH2POB_OK;
}

// adjoint of the transforms (sph2pob_transform_bwd_f32 / _general_f32)
int sph2pob_transform_bwd_f32_cpu(const float* b1, const float* b2, const float* g1, const float* g2, float* gb1, float* gb2, int64_t n,
                                  int box_dim, int variant, int edge, int jitter, void*) {
    int rc = check_common(box_dim, variant, edge, 0);
    if (rc) return rc;
    if ((variant & 0xff) > SPH2POB_VARIANT_EFFICIENT) return SPH2POB_ERR_OPTION;
    if (n < 0 || n > kMaxElems) return SPH2POB_ERR_SIZE;
    if (n == 0) return SPH2POB_OK;
    if (!b1 || !b2 || !g1 || !g2 || !gb1 || !gb2) return SPH2POB_ERR_NULL;
    auto body = [&](auto vtag, auto dtag) {
        constexpr int V = decltype(vtag)::value, D = decltype(dtag)::value;
        parallel_for(n, 1024, [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; i++) {
                float x[5], y[5], p[5], q[5], gx[5], gy[5];
                load_box<D>(b1, i, x);
                load_box<D>(b2, i, y);
                for (int k = 0; k < 5; k++) { p[k] = g1[i * 5 + k]; q[k] = g2[i * 5 + k]; }
                pair_transform_bwd<V, D>(x, y, p, q, edge, jitter != 0, gx, gy);
                for (int k = 0; k < D; k++) { gb1[i * D + k] = gx[k]; gb2[i * D + k] = gy[k]; }
            }
        });
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
    if ((variant & 0xff) == 0) { if (box_dim == 4) body(I0{}, I4{}); else body(I0{}, I5{}); }
    else { if (box_dim == 4) body(I1{}, I4{}); else body(I1{}, I5{}); }
    return SPH2POB_OK;
}

int sph2pob_transform_bwd_general_f32_cpu(const float* b1, const float* b2, const float* g1, const float* g2, float* gb1, float* gb2,
                                          int64_t n, int box_dim, int variant, int edge, int angle, int jitter, void*) {
    int rc = check_common(box_dim, variant, edge, angle);
    if (rc) return rc;
    if ((variant & 0xff) > SPH2POB_VARIANT_LEGACY) return SPH2POB_ERR_OPTION;
    if (n < 0 || n > kMaxElems) return SPH2POB_ERR_SIZE;
    if (n == 0) return SPH2POB_OK;
    if (!b1 || !b2 || !g1 || !g2 || !gb1 || !gb2) return SPH2POB_ERR_NULL;
    auto body = [&](auto vtag, auto dtag) {
        constexpr int V = decltype(vtag)::value, D = decltype(dtag)::value;
        parallel_for(n, 512, [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; i++) {
                float x[5], y[5], p[5], q[5], gx[5], gy[5];
                load_box<D>(b1, i, x);
                load_box<D>(b2, i, y);
                for (int k = 0; k < 5; k++) { p[k] = g1[i * 5 + k]; q[k] = g2[i * 5 + k]; }
                transform_bwd_dual<V, D>(x, y, p, q, edge, angle, jitter != 0, gx, gy);
                for (int k = 0; k < D; k++) { gb1[i * D + k] = gx[k]; gb2[i * D + k] = gy[k]; }
            }
        });
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
    const int v = variant & 0xff;
    if (v == 0) { if (box_dim == 4) body(I0{}, I4{}); else body(I0{}, I5{}); }
    else if (v == 1) { if (box_dim == 4) body(I1{}, I4{}); else body(I1{}, I5{}); }
    else body(I2{}, I4{});
    return SPH2POB_OK;
}

// ---- NMS on boxes sorted by (class, -score): sph2pob_nms_segmented_f32 / sph2pob_nms_f32 ----
int sph2pob_nms_segmented_f32_cpu(const float* boxes_sorted, const int64_t* cls_sorted, int64_t k, int box_dim, int variant_flags,
                                  float iou_threshold, int64_t max_segment, void* workspace, unsigned char* keep, void*) {
    (void)max_segment; (void)workspace;   // no suppression matrix on the host: no per-class limit either
    const int variant = variant_flags & 0xff;
    if (variant_flags & ~(0xff | SPH2POB_FLAG_REFERENCE_ORDER | SPH2POB_FLAG_ROBUST_PARALLEL | SPH2POB_FLAG_NAIVE_TAN)) return SPH2POB_ERR_OPTION;
    if ((variant_flags & SPH2POB_FLAG_NAIVE_TAN) && variant != SPH2POB_VARIANT_NAIVE) return SPH2POB_ERR_OPTION;
    if (box_dim != 4 && box_dim != 5) return SPH2POB_ERR_DIM;
    if (variant != SPH2POB_VARIANT_STANDARD && variant != SPH2POB_VARIANT_EFFICIENT && variant != SPH2POB_VARIANT_UNBIASED &&
        variant != SPH2POB_VARIANT_NAIVE)
        return SPH2POB_ERR_OPTION;
    if (k < 0) return SPH2POB_ERR_SIZE;
    if (k == 0) return SPH2POB_OK;
    if (!boxes_sorted || !keep) return SPH2POB_ERR_NULL;
    return dispatch(variant_flags, box_dim, NmsRun{boxes_sorted, cls_sorted, k, iou_threshold, keep,
                                                   (variant_flags & SPH2POB_FLAG_NAIVE_TAN) ? (int)EDGE_TANGENT : (int)EDGE_ARC});
}
int sph2pob_nms_f32_cpu(const float* boxes_sorted, const int64_t* cls_sorted, int64_t k, int box_dim, int variant_flags, float iou_threshold,
                        void* workspace, unsigned char* keep, void* stream) {
    return sph2pob_nms_segmented_f32_cpu(boxes_sorted, cls_sorted, k, box_dim, variant_flags, iou_threshold, k, workspace, keep, stream);
}

// ---- MaxIoUAssigner epilogue on a (k, n) matrix: sph2pob_assign_f32 (mmdet max_iou_assigner.py:135-220) ----
int sph2pob_assign_f32_cpu(const float* ov, int64_t k, int64_t n, float pos_iou_thr, float neg_iou_lo, float neg_iou_hi, float min_pos_iou,
                           int match_low_quality, int gt_max_assign_all, const int64_t* gt_labels, float* max_overlaps,
                           int64_t* argmax_overlaps, float* gt_max_overlaps, int64_t* gt_argmax_overlaps, int64_t* assigned_gt_inds,
                           int64_t* assigned_labels, void* workspace, void*) {
    (void)workspace;
    if (k <= 0 || n <= 0) return SPH2POB_ERR_SIZE;
    if (!ov || !max_overlaps || !argmax_overlaps || !gt_max_overlaps || !gt_argmax_overlaps || !assigned_gt_inds ||
        (assigned_labels && !gt_labels))
        return SPH2POB_ERR_NULL;
    parallel_for(n, 4096, [&](int64_t lo, int64_t hi) {   // columns: max / first argmax over the rows, thresholds
        for (int64_t j = lo; j < hi; j++) {
            float best = ov[j];
            int64_t bi = 0;
            for (int64_t i = 1; i < k; i++) {
                const float v = ov[i * n + j];
                if (v > best || (v != v && best == best)) { best = v; bi = i; }
            }
            max_overlaps[j] = best;
            argmax_overlaps[j] = bi;
            int64_t a = -1;
            if (best >= neg_iou_lo && best < neg_iou_hi) a = 0;
            if (best >= pos_iou_thr) a = bi + 1;
            assigned_gt_inds[j] = a;
        }
    });
    parallel_for(k, 1, [&](int64_t lo, int64_t hi) {      // rows: max / first argmax over the columns
        for (int64_t i = lo; i < hi; i++) {
            const float* row = ov + i * n;
            float best = row[0];
            int64_t bj = 0;
            for (int64_t j = 1; j < n; j++)
                if (row[j] > best || (row[j] != row[j] && best == best)) { best = row[j]; bj = j; }
            gt_max_overlaps[i] = best;
            gt_argmax_overlaps[i] = bj;
        }
    });
    if (match_low_quality) {   // later GTs overwrite earlier ones, like the reference's loop
        for (int64_t i = 0; i < k; i++) {
            const float g = gt_max_overlaps[i];
            if (!(g >= min_pos_iou)) continue;
            if (gt_max_assign_all) {
                const float* row = ov + i * n;
                for (int64_t j = 0; j < n; j++)
                    if (row[j] == g) assigned_gt_inds[j] = i + 1;
            } else {
                assigned_gt_inds[gt_argmax_overlaps[i]] = i + 1;
            }
        }
    }
    if (assigned_labels)
        for (int64_t j = 0; j < n; j++) assigned_labels[j] = assigned_gt_inds[j] > 0 ? gt_labels[assigned_gt_inds[j] - 1] : -1;
    return SPH2POB_OK;
}

}  // extern "C"

// ---- box coders and the OBB L1 loss body (sph2pob_coder.hpp: the rows the kernels of sph2pob_coder.hip compute) ----
namespace {
namespace C = sph2pob_coder;

template <int DIM>
void coder_encode_rows(const float* proposals, const float* gt, const C::Norm& nm, float* deltas, int64_t n) {
    parallel_for(n, 1 << 14, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; i++) {
            float d[5];
            C::encode_one<DIM>(proposals + i * DIM, gt + i * DIM, nm, d);
            for (int k = 0; k < DIM; k++) deltas[i * DIM + k] = d[k];
        }
    });
}

template <int DIM, bool BWD>
void coder_decode_rows(const float* rois, const float* deltas, const float* grad_boxes, const C::Norm& nm, float* out, int64_t total,
                       int num_classes, float max_ratio, int flags, float ctr_clamp) {
    parallel_for(total, 1 << 14, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; i++) {
            float p[5] = {0, 0, 0, 0, 0}, b[5], j[5];
            const float* r = rois + (num_classes == 1 ? i : i / num_classes) * DIM;
            for (int k = 0; k < DIM; k++) p[k] = r[k];
            C::decode_one<DIM, BWD>(p, deltas + i * DIM, nm, max_ratio, flags, ctr_clamp, b, BWD ? j : nullptr);
            for (int k = 0; k < DIM; k++) out[i * DIM + k] = BWD ? grad_boxes[i * DIM + k] * j[k] : b[k];
        }
    });
}
}  // namespace

extern "C" {

int sph2pob_coder_encode_f32_cpu(const float* proposals, const float* gt, const float* means_host, const float* stds_host,
                                 float* deltas, int64_t n, int box_dim, void*) {
    if (int rc = C::check_encode(proposals, gt, deltas, n, box_dim)) return rc;
    if (n == 0) return 0;
    const C::Norm nm = C::make_norm(means_host, stds_host, box_dim);
    if (box_dim == 4) coder_encode_rows<4>(proposals, gt, nm, deltas, n);
    else coder_encode_rows<5>(proposals, gt, nm, deltas, n);
    return SPH2POB_OK;
}

int sph2pob_coder_decode_f32_cpu(const float* rois, const float* deltas, const float* means_host, const float* stds_host,
                                 float* boxes, int64_t n, int num_classes, int box_dim, float max_ratio, int flags,
                                 float ctr_clamp, void*) {
    if (int rc = C::check_decode(rois, deltas, deltas, boxes, n, num_classes, box_dim, max_ratio, flags)) return rc;
    if (n == 0) return 0;
    const C::Norm nm = C::make_norm(means_host, stds_host, box_dim);
    if (box_dim == 4) coder_decode_rows<4, false>(rois, deltas, nullptr, nm, boxes, n * num_classes, num_classes, max_ratio, flags, ctr_clamp);
    else coder_decode_rows<5, false>(rois, deltas, nullptr, nm, boxes, n * num_classes, num_classes, max_ratio, flags, ctr_clamp);
    return SPH2POB_OK;
}

int sph2pob_coder_decode_bwd_f32_cpu(const float* rois, const float* deltas, const float* grad_boxes, const float* means_host,
                                     const float* stds_host, float* grad_deltas, int64_t n, int num_classes, int box_dim,
                                     float max_ratio, int flags, float ctr_clamp, void*) {
    if (int rc = C::check_decode(rois, deltas, grad_boxes, grad_deltas, n, num_classes, box_dim, max_ratio, flags)) return rc;
    if (n == 0) return 0;
    const C::Norm nm = C::make_norm(means_host, stds_host, box_dim);
    if (box_dim == 4) coder_decode_rows<4, true>(rois, deltas, grad_boxes, nm, grad_deltas, n * num_classes, num_classes, max_ratio, flags, ctr_clamp);
    else coder_decode_rows<5, true>(rois, deltas, grad_boxes, nm, grad_deltas, n * num_classes, num_classes, max_ratio, flags, ctr_clamp);
    return SPH2POB_OK;
}

int sph2pob_obb_l1_fwd_f32_cpu(const float* planar_pred, const float* planar_target, const float* weight, float scale, float* loss,
                               int64_t n, int flags, void*) {
    if (int rc = C::check_l1(planar_pred, planar_target, planar_pred, loss, n, flags)) return rc;
    if (n == 0) return 0;
    parallel_for(n, 1 << 14, [&](int64_t lo, int64_t hi) {
        const float ones[5] = {1.0f, 1.0f, 1.0f, 1.0f, 1.0f};
        for (int64_t i = lo; i < hi; i++) {
            float d[5];
            C::l1_fwd_one(planar_pred + i * 5, planar_target + i * 5, weight ? weight + i * 5 : ones, scale, flags, d);
            for (int k = 0; k < 5; k++) loss[i * 5 + k] = d[k];
        }
    });
    return SPH2POB_OK;
}

int sph2pob_obb_l1_bwd_f32_cpu(const float* planar_pred, const float* planar_target, const float* weight, const float* grad_loss,
                               float scale, float* grad_pred, float* grad_target, int64_t n, int flags, void*) {
    if (int rc = C::check_l1(planar_pred, planar_target, grad_loss, grad_pred, n, flags)) return rc;
    if (n == 0) return 0;
    parallel_for(n, 1 << 14, [&](int64_t lo, int64_t hi) {
        const float ones[5] = {1.0f, 1.0f, 1.0f, 1.0f, 1.0f};
        for (int64_t i = lo; i < hi; i++) {
            float u[5], ga[5], gb[5];
            for (int k = 0; k < 5; k++) u[k] = grad_loss[i * 5 + k];
            C::l1_bwd_one(planar_pred + i * 5, planar_target + i * 5, weight ? weight + i * 5 : ones, u, scale, flags, ga, gb);
            for (int k = 0; k < 5; k++) grad_pred[i * 5 + k] = ga[k];
            if (grad_target)
                for (int k = 0; k < 5; k++) grad_target[i * 5 + k] = gb[k];
        }
    });
    return SPH2POB_OK;
}

}  // extern "C"

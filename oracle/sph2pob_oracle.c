/*
 * ORACLE — test infrastructure only.  Plain-C CPU restatement of the reference's Sph2Pob hot path
 * (jitter -> Sph2Pob transform -> jitter -> planar rotated IoU -> clamp; OBB IoU/GIoU/DIoU/CIoU loss value).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (sph_retina_amd/) never does.  Parity status: PINNED against fixtures generated in-container from the
 * unmodified reference Python (oracle/gen_goldens.py -> tests/golden/NAME.npz) and against the reference's own
 * hard-coded sample pairs (tests/test_all_ious.py:244-261).  The mmcv-full 1.6.0 CUDA/C++ planar kernel is
 * restated from its published algorithm and pinned only through the reference's own assertion
 * mean|box_iou_rotated - diff_iou_rotated_2d| < 1e-6 (tests/test_sph_iou_loss.py:34).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "sph2pob_oracle.h"

/* ---- fp32 instantiation: the reference's arithmetic ---- */
#define REAL float
#define FN(x) x##_f32
#define MSIN sinf
#define MCOS cosf
#define MTAN tanf
#define MACOS acosf
#define MASIN asinf
#define MATAN atanf
#define MATAN2 atan2f
#define MSQRT sqrtf
#define MFMOD fmodf
#include "sph2pob_oracle_impl.h"
#undef REAL
#undef FN
#undef MSIN
#undef MCOS
#undef MTAN
#undef MACOS
#undef MASIN
#undef MATAN
#undef MATAN2
#undef MSQRT
#undef MFMOD

/* ---- fp64 instantiation: accuracy truth ---- */
#define REAL double
#define FN(x) x##_f64
#define MSIN sin
#define MCOS cos
#define MTAN tan
#define MACOS acos
#define MASIN asin
#define MATAN atan
#define MATAN2 atan2
#define MSQRT sqrt
#define MFMOD fmod
#include "sph2pob_oracle_impl.h"
#undef REAL
#undef FN

#include "unbiased_iou_oracle.h"

int sph2pob_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

#define EXPORT_PAIR(T, S)                                                                                          \
    int sph2pob_oracle_iou_aligned_##S(const T* b1, const T* b2, T* out, int64_t n, int dim, int variant, int mode, \
                                       int edge, int angle, int planar, int nthreads) {                            \
        return iou_aligned_##S(b1, b2, out, n, dim, variant, mode, edge, angle, planar, nthreads);                 \
    }                                                                                                              \
    int sph2pob_oracle_iou_pairwise_##S(const T* b1, int64_t m, const T* b2, int64_t n, T* out, int dim,           \
                                        int variant, int mode, int edge, int angle, int planar, int nthreads) {    \
        return iou_pairwise_##S(b1, m, b2, n, out, dim, variant, mode, edge, angle, planar, nthreads);             \
    }                                                                                                              \
    int sph2pob_oracle_transform_##S(const T* b1, const T* b2, T* o1, T* o2, int64_t n, int dim, int variant,      \
                                     int edge, int angle, int jitter) {                                            \
        return transform_batch_##S(b1, b2, o1, o2, n, dim, variant, edge, angle, jitter);                          \
    }                                                                                                              \
    int sph2pob_oracle_planar_iou_##S(const T* p1, const T* p2, T* out, int64_t n, int mode, int planar) {         \
        return planar_batch_##S(p1, p2, out, n, mode, planar);                                                     \
    }                                                                                                              \
    int sph2pob_oracle_loss_##S(const T* pred, const T* target, T* loss, T* iou, int64_t n, int dim,               \
                                int loss_mode, double eps, int nthreads) {                                         \
        return loss_batch_##S(pred, target, loss, iou, n, dim, loss_mode, eps, nthreads);                          \
    }

EXPORT_PAIR(float, f32)
EXPORT_PAIR(double, f64)

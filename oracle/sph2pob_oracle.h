/* ORACLE (test infrastructure) option codes.  Numerically equal to include/sph2pob_hip.h on purpose,
 * but deliberately a separate header: the product never includes anything from oracle/. */
#ifndef SPH2POB_ORACLE_H
#define SPH2POB_ORACLE_H
#include <stdint.h>

enum { SPH2POB_VARIANT_STANDARD = 0, SPH2POB_VARIANT_EFFICIENT = 1, SPH2POB_VARIANT_LEGACY = 2,
       SPH2POB_VARIANT_SPH_IOU = 3, SPH2POB_VARIANT_FOV_IOU = 4 };
enum { SPH2POB_MODE_IOU = 0, SPH2POB_MODE_IOF = 1 };
enum { SPH2POB_EDGE_ARC = 0, SPH2POB_EDGE_CHORD = 1, SPH2POB_EDGE_TANGENT = 2 };
enum { SPH2POB_ANGLE_EQUATOR = 0, SPH2POB_ANGLE_PROJECT = 1 };
enum { SPH2POB_PLANAR_MMCV = 0, SPH2POB_PLANAR_DIFF = 1, SPH2POB_PLANAR_EXACT = 2 };
enum { SPH2POB_LOSS_IOU = 0, SPH2POB_LOSS_GIOU = 1, SPH2POB_LOSS_DIOU = 2, SPH2POB_LOSS_CIOU = 3 };
enum { SPH2POB_ERR_DIM = -2 };

#ifdef __cplusplus
extern "C" {
#endif
int sph2pob_oracle_max_threads(void);
#define SPH2POB_ORACLE_DECL(T, S)                                                                                    \
    int sph2pob_oracle_iou_aligned_##S(const T* b1, const T* b2, T* out, int64_t n, int dim, int variant, int mode,  \
                                       int edge, int angle, int planar, int nthreads);                              \
    int sph2pob_oracle_iou_pairwise_##S(const T* b1, int64_t m, const T* b2, int64_t n, T* out, int dim,            \
                                        int variant, int mode, int edge, int angle, int planar, int nthreads);      \
    int sph2pob_oracle_transform_##S(const T* b1, const T* b2, T* o1, T* o2, int64_t n, int dim, int variant,       \
                                     int edge, int angle, int jitter);                                              \
    int sph2pob_oracle_planar_iou_##S(const T* p1, const T* p2, T* out, int64_t n, int mode, int planar);           \
    int sph2pob_oracle_loss_##S(const T* pred, const T* target, T* loss, T* iou, int64_t n, int dim, int loss_mode, \
                                double eps, int nthreads);
SPH2POB_ORACLE_DECL(float, f32)
SPH2POB_ORACLE_DECL(double, f64)
/* Unbiased IoU (sphdet/iou/unbiased_iou_{bfov,rbfov}.py + sph_iou_api.py:103-126); prec: see unbiased_iou_oracle.h */
int sph2pob_oracle_unbiased_iou(const double* b1, const double* b2, double* out, int64_t n, int dim, int prec,
                                int nthreads);
#ifdef __cplusplus
}
#endif
#endif

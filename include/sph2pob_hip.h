/*
 * sph2pob_hip.h — C ABI of libsph2pob_hip.so: the MI355X (gfx950) Sph2Pob spherical-IoU engine.
 *
 * The reference (ManuelVeras/sph-retina) has no FFI of its own: its hot path is Python that calls torch
 * element-wise ops plus three compiled mmcv-full 1.6.0 ops.  Each entry point below names the reference
 * interface it replaces (paths relative to the reference checkout).  A maintainer binds these from Python
 * with ctypes; no torch types cross this boundary.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless stated; row-major, contiguous, fp32 ("f32") / int64
 *   - boxes are degrees: BFoV (theta, phi, alpha, beta) box_dim = 4; RBFoV (+gamma) box_dim = 5
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); launchers only ENQUEUE:
 *     no allocation, no synchronisation, no ownership transfer, inputs are never written
 *   - return value: 0 on success, a negative SPH2POB_ERR_* for bad arguments, or a positive hipError_t
 */
#ifndef SPH2POB_HIP_H
#define SPH2POB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* transform variant: sphdet/iou/sph_iou_api.py:91-98 (sph2pob_{standard,efficient,legacy}_iou) */
enum { SPH2POB_VARIANT_STANDARD = 0, SPH2POB_VARIANT_EFFICIENT = 1, SPH2POB_VARIANT_LEGACY = 2,
       /* the cheap approximate backends of SphOverlaps2D (sphdet/iou/sph_iou_api.py:128-175; BFoV, mode 'iou' only):
        * Sph-IoU and FoV-IoU closed forms of sphdet/iou/approximate_ious.py:3-54 — SURVEY §8f-4 */
       SPH2POB_VARIANT_SPH_IOU = 3, SPH2POB_VARIANT_FOV_IOU = 4,
       /* Unbiased IoU: exact spherical-polygon intersection area, BFoV and RBFoV, mode 'iou' only —
        * sphdet/iou/sph_iou_api.py:103-126 over unbiased_iou_bfov.py:4-204 / unbiased_iou_rbfov.py:4-182 (numpy on the
        * CPU in the reference; the default backend of SphOverlaps2D and an SphNMS calculator, sph_nms.py:11-12).
        * Double precision on the device.  With SPH2POB_FLAG_REFERENCE_ORDER it also reproduces the fp32 roundings
        * numpy applies to fp32 inputs (noisier: the reference's fp32 areas cancel catastrophically on small boxes). */
       SPH2POB_VARIANT_UNBIASED = 5,
       /* Naive IoU: planar IoU of the boxes in ERP pixel space (sph_iou_api.py:179-197; sph_nms.py:13-14), BFoV via
        * axis-aligned boxes, RBFoV via rotated boxes; no jitter, mode 'iou' only */
       SPH2POB_VARIANT_NAIVE = 6 };
/* OR-ed into `variant`: evaluate the transform in the reference's own fp32 operation order (bit-for-bit the
 * arithmetic of sph2pob_standard.py / sph2pob_efficient.py, ~3x the VALU work) instead of the closed-form core.
 * Both meet the parity bar on the benchmark distribution; on close-centre pairs the closed-form core is ~10x
 * closer to fp64 truth, the reference-order path ~3x closer to the reference's own fp32 rounding. */
enum { SPH2POB_FLAG_REFERENCE_ORDER = 0x100,
       /* accepted and ignored: round 1's opt-in near-parallel safeguard (pairs whose planar boxes the reference's two jitter
        * steps leave parallel to < 2.5e-4 rad take a first-order form without 1/sin(delta)) is part of every closed-form
        * kernel since round 2 */
       SPH2POB_FLAG_ROBUST_PARALLEL = 0x200,
       /* SPH2POB_VARIANT_NAIVE only: Sph2PlanarBoxTransform('sph2tan') instead of 'sph2pix' (naive_iou(box_formator=...),
        * sphdet/iou/sph_iou_api.py:179, sphdet/bbox/box_formator.py:98-106, :166-172) */
       SPH2POB_FLAG_NAIVE_TAN = 0x400 };
/* mode: sphdet/iou/sph_iou_api.py:49 ('iou' | 'iof') */
enum { SPH2POB_MODE_IOU = 0, SPH2POB_MODE_IOF = 1 };
/* rbb_edge: sphdet/iou/sph2pob_standard.py:110-118 */
enum { SPH2POB_EDGE_ARC = 0, SPH2POB_EDGE_CHORD = 1, SPH2POB_EDGE_TANGENT = 2 };
/* rbb_angle: sphdet/iou/sph2pob_standard.py:88-108 */
enum { SPH2POB_ANGLE_EQUATOR = 0, SPH2POB_ANGLE_PROJECT = 1 };
/* loss mode: sphdet/losses/sph2pob_iou_loss.py:19 */
enum { SPH2POB_LOSS_IOU = 0, SPH2POB_LOSS_GIOU = 1, SPH2POB_LOSS_DIOU = 2, SPH2POB_LOSS_CIOU = 3 };

enum {
    SPH2POB_OK = 0,
    SPH2POB_ERR_NULL = -1,    /* a required pointer is NULL while the element count is > 0 */
    SPH2POB_ERR_DIM = -2,     /* box_dim not in {4, 5}, or legacy variant with box_dim 5 (reference raises) */
    SPH2POB_ERR_OPTION = -3,  /* variant / mode / edge / angle / loss mode out of range */
    SPH2POB_ERR_SIZE = -4     /* negative count or a product that overflows the launch geometry */
};

/* Library identification: ABI version (bumped on any signature change) and the code-object target. */
int sph2pob_abi_version(void);
const char* sph2pob_target_arch(void);
const char* sph2pob_error_string(int code);

/*
 * Aligned Sph2Pob IoU: out[i] = clamp(IoU(b1[i], b2[i]), 0, 1), i < n.
 * Replaces _sph2pob_iou_auxiliary(..., is_aligned=True) = jitter -> transform -> jitter -> mmcv
 * box_iou_rotated -> clamp: sphdet/iou/sph_iou_api.py:48-86 (and the wrappers at :91-98).
 * b1, b2: (n, box_dim) f32; out: (n) f32.  Algorithmic HBM traffic: 2*4*box_dim + 4 bytes per pair.
 */
int sph2pob_iou_aligned_f32(const float* b1, const float* b2, float* out, int64_t n, int box_dim, int variant,
                            int mode, int edge, int angle, void* stream);

/*
 * Pairwise Sph2Pob IoU: out[i*n + j] = clamp(IoU(b1[i], b2[j]), 0, 1); rows = first argument, exactly the
 * (rows, cols) view of sphdet/iou/sph_iou_api.py:59-64,85 without materialising the m*n expanded pairs.
 * This is the call MaxIoUAssigner makes: overlaps = iou_calculator(gt_bboxes, bboxes)
 * (mmdet/core/bbox/assigners/max_iou_assigner.py:113).
 */
int sph2pob_iou_pairwise_f32(const float* b1, int64_t m, const float* b2, int64_t n, float* out, int box_dim,
                             int variant, int mode, int edge, int angle, void* stream);

/*
 * Planar oriented boxes of both roles, (n, 5) f32 each = (x, y, w, h, a[rad]); with jitter != 0 the spherical
 * and rotated jitters are applied around the transform exactly as Sph2PobTransfrom.new_forward does
 * (sphdet/losses/sph2pob_transform.py:26-30) — the shared front end of every Sph2Pob-wrapped OBB loss.
 * Replaces sph2pob_{standard,efficient,legacy}(sph_gt, sph_pred, rbb_angle_version='rad', ...):
 * sphdet/iou/sph2pob_standard.py:8-80, sph2pob_efficient.py:9-73, sph2pob_legacy.py:8-31.
 */
int sph2pob_transform_f32(const float* b1, const float* b2, float* planar1, float* planar2, int64_t n,
                          int box_dim, int variant, int edge, int angle, int jitter, void* stream);

/*
 * Adjoint of sph2pob_transform_f32 for variant STANDARD | EFFICIENT, rbb_angle 'equator': given the gradients of a
 * scalar w.r.t. the two (n, 5) planar boxes, writes its gradients w.r.t. the two (n, box_dim) spherical boxes
 * (degrees).  This is what lets every Sph2Pob-wrapped OBB loss (Sph2PobTransfrom.new_forward,
 * sphdet/losses/sph2pob_transform.py:24-35: L1 / GD / KF / IoU bodies) back-propagate to the spherical inputs without
 * torch autograd through the ~70 transform ops.  With jitter != 0 the clamp gates of both jitters are applied.
 */
int sph2pob_transform_bwd_f32(const float* b1, const float* b2, const float* grad_planar1, const float* grad_planar2,
                              float* grad_b1, float* grad_b2, int64_t n, int box_dim, int variant, int edge, int jitter,
                              void* stream);

/*
 * The same adjoint for the transforms without a closed-form backward here — sph2pob_legacy (BFoV only) and
 * rbb_angle = 'project' of sph2pob_standard / sph2pob_efficient (sphdet/iou/sph2pob_legacy.py:8-31,
 * sph2pob_standard.py:88-108, sph2pob_efficient.py:81-97: the reference differentiates them with torch autograd) — by
 * forward-mode differentiation of the reference-order transform (2 * box_dim passes on (value, derivative) pairs).
 * jitter != 0: the adjoint of jitter_spherical -> transform -> jitter_rotated (Sph2PobTransfrom('sph2pob_legacy'),
 * sphdet/losses/sph2pob_transform.py:12-16, :28-30).
 */
int sph2pob_transform_bwd_general_f32(const float* b1, const float* b2, const float* grad_planar1,
                                      const float* grad_planar2, float* grad_b1, float* grad_b2, int64_t n, int box_dim,
                                      int variant, int edge, int angle, int jitter, void* stream);

/*
 * Planar rotated-rectangle IoU on GIVEN planar boxes (x, y, w, h, a [rad]) — the op the reference obtains from mmcv
 * (`mmcv.ops.box_iou_rotated`, call sites sphdet/iou/sph_iou_api.py:79, :193; the vendored value-equivalent
 * `diff_iou_rotated_2d`, sphdet/iou/diff_iou_rotated.py:325-343, called directly by tests/test_all_ious.py:22-24).
 * aligned != 0: out[i] = IoU(p1[i], p2[i]), m == n; aligned == 0: out[i * n + j] = IoU(p1[i], p2[j]) (rows = p1).
 * mode: SPH2POB_MODE_IOU | SPH2POB_MODE_IOF.  No jitter and no clamp (as the mmcv op): exactly parallel edges are
 * handled by the boundary integral's clamped reciprocals, nearly parallel ones (|sin| < 2.5e-4) in double.
 */
int sph2pob_planar_iou_f32(const float* p1, int64_t m, const float* p2, int64_t n, float* out, int aligned, int mode,
                           void* stream);

/*
 * Sph2PobIoULoss element values: loss[i] = scale * w_i * L(pred[i], target[i]), L = 1 - IoU | GIoU | DIoU | CIoU
 * form; scale carries loss_weight (sph2pob_iou_loss.py:49).
 * Replaces Sph2PobTransfrom.new_forward (sphdet/losses/sph2pob_transform.py:24-35: clone, spherical jitter,
 * sph2pob_standard(..., 'rad'), rotated jitter) + obb_iou_loss (sphdet/losses/sph2pob_iou_loss.py:104-196, whose
 * IoU is mmcv diff_iou_rotated_2d, :122) + the element-weight step of weight_reduce_loss
 * (mmdet/models/losses/utils.py:44-45).  weight: NULL, (n) [weight_dim 1] or (n, box_dim) [weight_dim box_dim:
 * the per-box mean is taken as OBBIoULoss.forward does, sph2pob_iou_loss.py:43-48].  iou (optional, may be NULL)
 * receives the clamped planar IoU.  eps = the loss's eps (default 1e-6, sph2pob_iou_loss.py:17).
 */
int sph2pob_loss_fwd_f32(const float* pred, const float* target, const float* weight, int weight_dim, float scale,
                         float* loss, float* iou, int64_t n, int box_dim, int loss_mode, float eps, void* stream);

/*
 * Adjoint of the above w.r.t. the spherical inputs (degrees): grad_pred[i,:] = g_i * dL_i/dpred[i,:], likewise
 * grad_target (optional, may be NULL), with g_i = grad_out[i * grad_stride] * scale * w_i  (grad_stride 0 =
 * one scalar upstream gradient, e.g. of a 'mean'/'sum' reduced loss; 1 = per element, reduction 'none').
 * Replaces torch autograd through the ~150 ops the reference records for this loss.  Recomputes the forward in
 * registers: reads only pred, target, weight and grad_out.
 */
int sph2pob_loss_bwd_f32(const float* pred, const float* target, const float* weight, int weight_dim,
                         const float* grad_out, int grad_stride, float scale, float* grad_pred, float* grad_target,
                         int64_t n, int box_dim, int loss_mode, float eps, void* stream);

/*
 * The same loss REDUCED in one go: out[0] = scale * sum_i w_i * L(pred[i], target[i]) — the forward kernel leaves one
 * partial sum per workgroup in `workspace` (no element buffer is written or re-read), a second launch adds the partials
 * in a fixed order (bitwise reproducible, no float atomics).  scale carries loss_weight and the 1 / n | 1 / (avg_factor +
 * eps) of weight_reduce_loss (mmdet/models/losses/utils.py:47-58).  workspace: device buffer of at least
 * sph2pob_loss_sum_workspace_floats(n) floats.  Backward: sph2pob_loss_bwd_f32 with grad_stride 0 and the same scale.
 */
int64_t sph2pob_loss_sum_workspace_floats(int64_t n);
int sph2pob_loss_fwd_sum_f32(const float* pred, const float* target, const float* weight, int weight_dim, float scale,
                             float* out, float* workspace, int64_t n, int box_dim, int loss_mode, float eps,
                             void* stream);

/*
 * Forward AND gradients in one pass (the training call: the backward kernel recomputes the whole forward, so when the
 * gradient will be asked for anyway the loss value is a by-product of it): writes the loss elements (loss, may be NULL)
 * and / or their sum (out_sum + workspace as in sph2pob_loss_fwd_sum_f32, may be NULL), and
 *     grad_pred[i, :]   = scale * w_i * dL_i / dpred[i, :]        (n, box_dim)
 *     grad_target[i, :] = scale * w_i * dL_i / dtarget[i, :]      (optional, may be NULL)
 * i.e. the gradients for an upstream gradient of 1.  torch's backward then only scales them:
 * sph2pob_loss_grad_scale_f32: out[i, :] = stash[i, :] * grad_out[i * grad_stride]  (grad_stride 0: one scalar).
 * out may be the stash itself (in place); in place with a scalar upstream gradient of exactly 1.0 — a plain
 * `loss.backward()` — the launch returns after one scalar load per workgroup: the stash already is the gradient.
 */
int sph2pob_loss_fwd_grad_f32(const float* pred, const float* target, const float* weight, int weight_dim, float scale,
                              float* loss, float* out_sum, float* workspace, float* grad_pred, float* grad_target,
                              int64_t n, int box_dim, int loss_mode, float eps, void* stream);
int sph2pob_loss_grad_scale_f32(const float* stash, const float* grad_out, int grad_stride, float* out, int64_t n,
                                int box_dim, void* stream);

/*
 * out[0] = scale * sum(x[0..n)) — deterministic two-pass tree (bitwise reproducible, no float atomics); the
 * reduction step of weight_reduce_loss (mmdet/models/losses/utils.py:47-55).  workspace: device buffer of at
 * least sph2pob_sum_workspace_floats() floats.
 */
int sph2pob_sum_workspace_floats(void);
int sph2pob_sum_f32(const float* x, int64_t n, float scale, float* out, float* workspace, void* stream);

/*
 * Greedy per-class NMS with the Sph2Pob IoU as overlap: keep[i] = 1 iff sorted box i survives.
 * Replaces sph_nms_op (sphdet/bbox/nms/sph_nms.py:62-74: python while-loop, one full IoU pipeline + host sync per
 * kept box) for every class at once.  Input boxes must be sorted by (class ascending, score descending);
 * cls_sorted may be NULL (class-agnostic).  A box j is suppressed by an earlier kept box i of the same class when
 * IoU(box_i as bboxes1, box_j as bboxes2) > iou_threshold (`iou <= thr` keeps, :72).  variant: EFFICIENT (what
 * SphNMS('sph2pob_efficient') uses, sph_nms.py:9-10) or STANDARD.  k <= sph2pob_nms_max_boxes();
 * workspace: device buffer of sph2pob_nms_workspace_bytes(k) bytes (the k x ceil(k/64) suppression bit-matrix).
 */
int sph2pob_nms_max_boxes(void);
int64_t sph2pob_nms_workspace_bytes(int64_t k);
int sph2pob_nms_f32(const float* boxes_sorted, const int64_t* cls_sorted, int64_t k, int box_dim, int variant,
                    float iou_threshold, void* workspace, unsigned char* keep, void* stream);
/* Same with a caller-supplied bound on the largest class segment (boxes of one class): the suppression matrix is then
 * k x (max_segment / 64 + 2) words instead of k x k / 64, and k itself is unbounded (< 2^31); max_segment <=
 * sph2pob_nms_max_boxes() (32 704).  cls_sorted == NULL means one segment (max_segment >= k).  Under-stating max_segment
 * truncates suppression for the over-long segment (never an out-of-bounds access).  multiclass_nms hands ALL
 * (box, class) candidates above score_thr to the NMS (sphdet/bbox/nms/utils.py:6-15): 5 000 boxes x 37 classes is
 * beyond the one-matrix limit of sph2pob_nms_f32 but ~1 700 per class. */
int64_t sph2pob_nms_segmented_workspace_bytes(int64_t k, int64_t max_segment);
int sph2pob_nms_segmented_f32(const float* boxes_sorted, const int64_t* cls_sorted, int64_t k, int box_dim, int variant,
                              float iou_threshold, int64_t max_segment, void* workspace, unsigned char* keep,
                              void* stream);

/*
 * sph_batched_nms without the host (sphdet/bbox/nms/sph_nms.py:22-60): UNSORTED boxes (k, box_dim), scores (k), class ids
 * idxs (k, int64; NULL = one class: sph_nms_op, :62-74) -> the kept boxes' original indices in descending-score order
 * (ties by ascending index), at most max_num of them, and dets = (box, score) rows of box_dim + 1 floats.  Four launches,
 * no host work: a one-workgroup bitonic sort of composite (class | -score | index) keys in LDS, the suppression matrix,
 * the per-class sweeps, and a one-workgroup sort of the kept boxes by score.  *status (device int) = number of rows written
 * to keep / dets, or -1 when a class id is outside [0, 262 143] (nothing valid was written: use the sorted-input entry
 * points).  k <= sph2pob_batched_nms_max_boxes() (16 384), any number per class.  keep / dets must hold min(max_num, k) rows;
 * workspace: sph2pob_batched_nms_workspace_bytes(k, box_dim) bytes, no initialisation.
 */
int sph2pob_batched_nms_max_boxes(void);
int64_t sph2pob_batched_nms_workspace_bytes(int64_t k, int box_dim);
int sph2pob_batched_nms_f32(const float* boxes, const float* scores, const int64_t* idxs, int64_t k, int box_dim, int variant,
                            float iou_threshold, int64_t max_num, void* workspace, int64_t* keep, float* dets, int* status,
                            void* stream);

/*
 * MaxIoUAssigner epilogue on a (k, n) overlaps matrix (rows = GT, columns = boxes), SURVEY §8f-1.
 * Replaces assign_wrt_overlaps (mmdet/core/bbox/assigners/max_iou_assigner.py:135-220) for k > 0, n > 0:
 *   max_overlaps, argmax_overlaps       = overlaps.max(dim=0)   (:171)   first maximal index on ties
 *   gt_max_overlaps, gt_argmax_overlaps = overlaps.max(dim=1)   (:174)
 *   assigned_gt_inds: -1; 0 where neg_iou_lo <= max < neg_iou_hi (:178-185); argmax + 1 where max >= pos_iou_thr
 *   (:188-189); low-quality matching (:200-207) for i ascending — the reference's python loop costs one host sync
 *   per GT.  assigned_labels (optional) = gt_labels[gt_ind - 1] or -1 (:209-216).
 * workspace: sph2pob_assign_workspace_bytes(k, n) bytes.
 */
int64_t sph2pob_assign_workspace_bytes(int64_t k, int64_t n);
int sph2pob_assign_f32(const float* overlaps, int64_t k, int64_t n, float pos_iou_thr, float neg_iou_lo,
                       float neg_iou_hi, float min_pos_iou, int match_low_quality, int gt_max_assign_all,
                       const int64_t* gt_labels, float* max_overlaps, int64_t* argmax_overlaps, float* gt_max_overlaps,
                       int64_t* gt_argmax_overlaps, int64_t* assigned_gt_inds, int64_t* assigned_labels, void* workspace,
                       void* stream);

/*
 * MaxIoUAssigner.assign FUSED with the pairwise IoU (SURVEY §8f-1 as written: the k x n overlaps are never materialised).
 * Replaces `overlaps = self.iou_calculator(gt_bboxes, bboxes)` + `assign_wrt_overlaps`
 * (mmdet/core/bbox/assigners/max_iou_assigner.py:113, :135-220) for the closed-form sph2pob_standard_iou /
 * sph2pob_efficient_iou calculators (variant STANDARD | EFFICIENT, default arithmetic, rbb_angle 'equator', mode 'iou'):
 * the pairwise kernel keeps per-column and per-row running maxima (first index on ties, like torch.max) while it
 * finishes the pairs — culled pairs are exact zeros — and the finalize pass applies the thresholds and the low-quality
 * step (a GT row is re-evaluated only against a column tile that holds its maximum more than once).  Results are bit-identical to
 * sph2pob_iou_pairwise_f32 followed by sph2pob_assign_f32.
 *   ignore       optional (n) bytes: non-zero = the column's overlaps are -1 (`overlaps[:, ignore_max > thr] = -1`, :115-126)
 *   overlaps     optional (k, n): also write the matrix (for callers that want it; the assignment does not read it)
 *   gt_keys      (k) int64, the exchange format of the sharded form: per-GT (max IoU, first global column) as keys that
 *                order as SIGNED integers (torch.distributed / RCCL have no unsigned MAX)
 *   col_offset   global index of this shard's first column (0 when not sharded)
 *   workspace    sph2pob_iou_assign_workspace_bytes(k, n) bytes of scratch (per-chunk column partials, per-tile row
 *                partials): no initialisation, but untouched between reduce and finalize
 *   state        sph2pob_iou_assign_state_bytes(k, n) bytes (per-GT accumulators + arrival counters) that must be ZERO when a
 *                call is enqueued and are left zero by every completed call (one-call form: by the finalize pass; reduce: by
 *                its key pass): zero the buffer once after allocation, then reuse it stream-ordered (its layout moves with
 *                k: a buffer is clean for any (k, n) it is large enough for).  Kept apart from the scratch because the scratch's layout moves with k.
 * sph2pob_iou_assign_f32 = one device, two launches (pairwise kernel with the reductions; finalize).  The two halves exist
 * so that a job sharded on the box axis can put ONE all-reduce(MAX) of the k keys between them:
 * reduce = pairwise kernel + this shard's keys; finalize = column maxima, thresholds, low-quality step against the (global)
 * keys.  Outputs as for sph2pob_assign_f32 (argmax_overlaps, gt_max_overlaps, gt_argmax_overlaps, assigned_labels may be
 * NULL); gt_argmax_overlaps are global column indices.
 * Limits: k <= 262 140, n < 2^31 - 256, col_offset + n < 2^31 - 1.  Other variants / arithmetics: SPH2POB_ERR_OPTION (use
 * the two-call form on the matrix).
 */
int64_t sph2pob_iou_assign_workspace_bytes(int64_t k, int64_t n);
int64_t sph2pob_iou_assign_state_bytes(int64_t k, int64_t n);
int sph2pob_iou_assign_reduce_f32(const float* gt, int64_t k, const float* boxes, int64_t n, int box_dim, int variant, int edge,
                                  const unsigned char* ignore, int64_t col_offset, float* overlaps, int64_t* gt_keys,
                                  void* workspace, void* state, void* stream);
int sph2pob_iou_assign_finalize_f32(const float* gt, int64_t k, const float* boxes, int64_t n, int box_dim, int variant, int edge,
                                    int64_t col_offset, const int64_t* gt_keys, float pos_iou_thr, float neg_iou_lo,
                                    float neg_iou_hi, float min_pos_iou, int match_low_quality, int gt_max_assign_all,
                                    const int64_t* gt_labels, float* max_overlaps, int64_t* argmax_overlaps,
                                    float* gt_max_overlaps, int64_t* gt_argmax_overlaps, int64_t* assigned_gt_inds,
                                    int64_t* assigned_labels, void* workspace, void* stream);
int sph2pob_iou_assign_f32(const float* gt, int64_t k, const float* boxes, int64_t n, int box_dim, int variant, int edge,
                           const unsigned char* ignore, float* overlaps, float pos_iou_thr, float neg_iou_lo, float neg_iou_hi,
                           float min_pos_iou, int match_low_quality, int gt_max_assign_all, const int64_t* gt_labels,
                           float* max_overlaps, int64_t* argmax_overlaps, float* gt_max_overlaps, int64_t* gt_argmax_overlaps,
                           int64_t* assigned_gt_inds, int64_t* assigned_labels, void* workspace, void* state, void* stream);

/* ---- box coder: the step immediately in front of the loss when reg_decoded_bbox=True --------------
 * Replaces sphdet/bbox/coder/delta_xywh_sph_bbox_coder.py:116-161 (bbox2delta), :164-263 (delta2bbox) for box_dim 4
 * and sphdet/bbox/coder/delta_xywha_rsph_bbox_coder.py:116-164, :167-268 for box_dim 5 (fifth delta = deg2rad of the
 * gamma difference; decoded gamma clamped to [-90+1e-7, 90-1e-7]).
 *
 * means / stds: HOST pointers to box_dim floats (copied into the kernel arguments; NULL = zeros / ones).
 * encode : deltas[i] = ((gt - proposal) / size, log(size ratio)[, deg2rad(dgamma)] - means) / stds, widths clipped at 1e-7.
 * decode : rois (n, box_dim), deltas (n, num_classes*box_dim) -> boxes (n, num_classes*box_dim);
 *          flags: SPH2POB_CODER_CLIP_BORDER (clamp to the sphere ranges, the reference's clip_border=True),
 *                 SPH2POB_CODER_CTR_CLAMP  (add_ctr_clamp=True: centre shift clamped to +-ctr_clamp, dwh only from above);
 *          max_ratio = |log(wh_ratio_clip)|.
 * decode_bwd: grad_deltas = J^T grad_boxes with the clamp gates of decode (what autograd gives the reference when the
 *          decoded boxes feed Sph2PobIoULoss: sphdet/models/heads/sph_retina_head.py:255-264).
 */
enum { SPH2POB_CODER_CLIP_BORDER = 1, SPH2POB_CODER_CTR_CLAMP = 2 };
int sph2pob_coder_encode_f32(const float* proposals, const float* gt, const float* means_host, const float* stds_host,
                             float* deltas, int64_t n, int box_dim, void* stream);
int sph2pob_coder_decode_f32(const float* rois, const float* deltas, const float* means_host, const float* stds_host,
                             float* boxes, int64_t n, int num_classes, int box_dim, float max_ratio, int flags,
                             float ctr_clamp, void* stream);
int sph2pob_coder_decode_bwd_f32(const float* rois, const float* deltas, const float* grad_boxes,
                                 const float* means_host, const float* stds_host, float* grad_deltas, int64_t n,
                                 int num_classes, int box_dim, float max_ratio, int flags, float ctr_clamp,
                                 void* stream);

/* ---- OBB L1 loss body: Sph2PobL1Loss after the Sph2Pob transform ----------------------------------
 * Replaces sphdet/losses/sph2pob_l1_loss.py:28-88 on PLANAR boxes (x, y, w, h, a[rad]) as produced by
 * sph2pob_transform_f32(..., jitter=1):  loss[i,k] = scale * weight[i,k] * |d[i,k]| with
 *   ENCODE: d = bbox2delta(proposals, gt) = ((gx-px)/pw, (gy-py)/ph, log(gw/pw), log(gh/ph), (wrap(ga)-wrap(pa))/pi),
 *           widths clipped at 1e-7; proposals = pred, gt = target (SWAP: the other way round, :31-32);
 *           MODULUS: wrap(a) = (a + pi) mod pi (:84-88); the reference then takes L1 against zeros (:34);
 *   otherwise d = pred - target.
 * weight: (n,5) or NULL.  The backward gives the gradients w.r.t. both planar boxes (feed them to
 * sph2pob_transform_bwd_f32 to reach the spherical boxes).  grad_target may be NULL.
 */
enum { SPH2POB_L1_ENCODE = 1, SPH2POB_L1_SWAP = 2, SPH2POB_L1_MODULUS = 4 };
int sph2pob_obb_l1_fwd_f32(const float* planar_pred, const float* planar_target, const float* weight, float scale,
                           float* loss, int64_t n, int flags, void* stream);
int sph2pob_obb_l1_bwd_f32(const float* planar_pred, const float* planar_target, const float* weight,
                           const float* grad_loss, float scale, float* grad_pred, float* grad_target, int64_t n,
                           int flags, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPH2POB_HIP_H */

"""ORACLE — test infrastructure only (numpy + ctypes front end of oracle/sph2pob_oracle.c).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
The product package ``sph_retina_amd`` never imports anything from ``oracle/``.

Host-side pieces restated here in numpy (each cites the reference file:line it follows):
  * ``weight_reduce_loss``  — mmdet/models/losses/utils.py:30-59
  * ``OBBIoULoss.forward`` + ``Sph2PobTransfrom`` weight widening — sphdet/losses/sph2pob_iou_loss.py:25-58,
    sphdet/losses/sph2pob_transform.py:32-34
  * ``sph_nms_op`` / ``sph_batched_nms`` — sphdet/bbox/nms/sph_nms.py:22-74
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libsph2pob_oracle.so')

VARIANTS = {'standard': 0, 'efficient': 1, 'legacy': 2, 'sph_iou': 3, 'fov_iou': 4}
MODES = {'iou': 0, 'iof': 1}
EDGES = {'arc': 0, 'chord': 1, 'tangent': 2}
ANGLES = {'equator': 0, 'project': 1, None: 0}
PLANARS = {'mmcv': 0, 'diff': 1, 'exact': 2}
LOSS_MODES = {'iou': 0, 'giou': 1, 'diou': 2, 'ciou': 3}


def build(force=False):
    """Compile the C oracle (gcc).  Building the checker is not using it."""
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
            for f in ('sph2pob_oracle.c', 'sph2pob_oracle_impl.h', 'sph2pob_oracle.h', 'unbiased_iou_oracle.h')):
        subprocess.check_call(['make', '-C', _HERE, '-s'] + (['-B'] if force else []))
    return _SO


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        build()
        _LIB = ctypes.CDLL(_SO)
    return _LIB


def max_threads():
    return int(lib().sph2pob_oracle_max_threads())


def _np(a, dtype):
    return np.ascontiguousarray(np.asarray(a, dtype=dtype))


def _suffix(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return 'f32', ctypes.c_float
    if dtype == np.float64:
        return 'f64', ctypes.c_double
    raise TypeError(dtype)


def _ptr(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def iou_aligned(b1, b2, variant='standard', mode='iou', edge='arc', angle='equator', planar='mmcv',
                dtype=np.float32, nthreads=1):
    suf, ct = _suffix(dtype)
    b1, b2 = _np(b1, dtype), _np(b2, dtype)
    assert b1.shape == b2.shape and b1.ndim == 2
    n, dim = b1.shape
    out = np.empty(n, dtype=dtype)
    rc = getattr(lib(), 'sph2pob_oracle_iou_aligned_' + suf)(
        _ptr(b1, ct), _ptr(b2, ct), _ptr(out, ct), ctypes.c_int64(n), dim, VARIANTS[variant], MODES[mode],
        EDGES[edge], ANGLES[angle], PLANARS[planar], int(nthreads))
    if rc != 0:
        raise ValueError('oracle rc=%d' % rc)
    return out


def iou_pairwise(b1, b2, variant='standard', mode='iou', edge='arc', angle='equator', planar='mmcv',
                 dtype=np.float32, nthreads=1):
    suf, ct = _suffix(dtype)
    b1, b2 = _np(b1, dtype), _np(b2, dtype)
    m, dim = b1.shape
    n = b2.shape[0]
    assert b2.shape[1] == dim
    out = np.empty((m, n), dtype=dtype)
    rc = getattr(lib(), 'sph2pob_oracle_iou_pairwise_' + suf)(
        _ptr(b1, ct), ctypes.c_int64(m), _ptr(b2, ct), ctypes.c_int64(n), _ptr(out, ct), dim, VARIANTS[variant],
        MODES[mode], EDGES[edge], ANGLES[angle], PLANARS[planar], int(nthreads))
    if rc != 0:
        raise ValueError('oracle rc=%d' % rc)
    return out


def transform(b1, b2, variant='standard', edge='arc', angle='equator', jitter=False, dtype=np.float32):
    """Planar boxes (x, y, w, h, a[rad]) of both roles; with ``jitter`` both jitters are applied around it."""
    suf, ct = _suffix(dtype)
    b1, b2 = _np(b1, dtype), _np(b2, dtype)
    n, dim = b1.shape
    o1 = np.empty((n, 5), dtype=dtype)
    o2 = np.empty((n, 5), dtype=dtype)
    rc = getattr(lib(), 'sph2pob_oracle_transform_' + suf)(
        _ptr(b1, ct), _ptr(b2, ct), _ptr(o1, ct), _ptr(o2, ct), ctypes.c_int64(n), dim, VARIANTS[variant],
        EDGES[edge], ANGLES[angle], int(bool(jitter)))
    if rc != 0:
        raise ValueError('oracle rc=%d' % rc)
    return o1, o2


def planar_iou(p1, p2, mode='iou', planar='mmcv', dtype=np.float32):
    suf, ct = _suffix(dtype)
    p1, p2 = _np(p1, dtype), _np(p2, dtype)
    n = p1.shape[0]
    out = np.empty(n, dtype=dtype)
    getattr(lib(), 'sph2pob_oracle_planar_iou_' + suf)(_ptr(p1, ct), _ptr(p2, ct), _ptr(out, ct),
                                                         ctypes.c_int64(n), MODES[mode], PLANARS[planar])
    return out


def loss_elements(pred, target, mode='iou', eps=1e-6, dtype=np.float32, nthreads=1, return_iou=False):
    """Per-pair unweighted Sph2PobIoULoss element (sph2pob_transform.py:24-35 + sph2pob_iou_loss.py:104-196)."""
    suf, ct = _suffix(dtype)
    pred, target = _np(pred, dtype), _np(target, dtype)
    n, dim = pred.shape
    loss = np.empty(n, dtype=dtype)
    iou = np.empty(n, dtype=dtype)
    rc = getattr(lib(), 'sph2pob_oracle_loss_' + suf)(
        _ptr(pred, ct), _ptr(target, ct), _ptr(loss, ct), _ptr(iou, ct), ctypes.c_int64(n), dim, LOSS_MODES[mode],
        ctypes.c_double(eps), int(nthreads))
    if rc != 0:
        raise ValueError('oracle rc=%d' % rc)
    return (loss, iou) if return_iou else loss


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    """mmdet/models/losses/utils.py:30-59 (numpy, dtype of ``loss``)."""
    dt = loss.dtype
    if weight is not None:
        loss = loss * np.asarray(weight, dtype=dt)
    if avg_factor is None:
        if reduction == 'none':
            return loss
        if reduction == 'mean':
            return loss.mean(dtype=dt)
        if reduction == 'sum':
            return loss.sum(dtype=dt)
        raise ValueError(reduction)
    if reduction == 'mean':
        eps = np.finfo(np.float32).eps
        return loss.sum(dtype=dt) / dt.type(avg_factor + eps)
    if reduction != 'none':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


def sph2pob_iou_loss(pred, target, weight=None, avg_factor=None, mode='iou', eps=1e-6, reduction='mean',
                     loss_weight=1.0, dtype=np.float32):
    """Sph2PobIoULoss.forward value (sph2pob_transform.py:24-35 then sph2pob_iou_loss.py:25-58)."""
    pred, target = _np(pred, dtype), _np(target, dtype)
    box_version = target.shape[-1]
    if weight is not None:
        weight = _np(weight, dtype)
        if weight.ndim > 1 and box_version == 4:
            weight = np.concatenate([weight, weight.mean(-1, keepdims=True)], axis=-1)
        if not np.any(weight > 0):
            return np.dtype(dtype).type(0.0)
        if weight.ndim > 1:
            weight = weight.mean(-1)
    el = loss_elements(pred, target, mode=mode, eps=eps, dtype=dtype)
    return np.dtype(dtype).type(loss_weight) * weight_reduce_loss(el, weight, reduction, avg_factor)


def loss_grad_fd(pred, target, mode='iou', eps=1e-6, h=1e-5):
    """fp64 central finite differences of the per-pair loss element w.r.t. pred and target (degrees).
    Independent check for the hand-derived HIP adjoint; only meaningful away from the kinks of the loss."""
    pred = _np(pred, np.float64)
    target = _np(target, np.float64)
    n, dim = pred.shape
    gp = np.zeros_like(pred)
    gt = np.zeros_like(target)
    for k in range(dim):
        for arr, g in ((pred, gp), (target, gt)):
            save = arr[:, k].copy()
            arr[:, k] = save + h
            lp = loss_elements(pred, target, mode, eps, np.float64)
            arr[:, k] = save - h
            lm = loss_elements(pred, target, mode, eps, np.float64)
            arr[:, k] = save
            g[:, k] = (lp - lm) / (2 * h)
    return gp, gt


def transform_vjp_fd(b1, b2, g1, g2, variant='standard', edge='arc', jitter=True, h=1e-5):
    """f64 central finite differences of  sum(g1 * planar1 + g2 * planar2)  w.r.t. the spherical inputs (degrees):
    the vector-Jacobian product the HIP transform adjoint must reproduce."""
    b1 = _np(b1, np.float64)
    b2 = _np(b2, np.float64)
    g1 = _np(g1, np.float64)
    g2 = _np(g2, np.float64)

    def f():
        p1, p2 = transform(b1, b2, variant=variant, edge=edge, jitter=jitter, dtype=np.float64)
        return (g1 * p1).sum(1) + (g2 * p2).sum(1)
    out = []
    for arr in (b1, b2):
        g = np.zeros_like(arr)
        for k in range(arr.shape[1]):
            save = arr[:, k].copy()
            arr[:, k] = save + h
            fp = f()
            arr[:, k] = save - h
            fm = f()
            arr[:, k] = save
            g[:, k] = (fp - fm) / (2 * h)
        out.append(g)
    return out


def nms_op(boxes, scores, iou_threshold, variant='efficient', planar='mmcv', nthreads=1):
    """sph_nms_op: sphdet/bbox/nms/sph_nms.py:62-74.  Stable descending sort (ties keep input order)."""
    boxes = _np(boxes, np.float32)
    order = np.argsort(-np.asarray(scores, dtype=np.float32), kind='stable')
    keep = []
    while order.size > 0:
        keep.append(order[0])
        if order.size == 1:
            break
        if variant == 'unbiased':
            iou = unbiased_iou(boxes[order[:1]], boxes[order[1:]], is_aligned=False).reshape(-1)
        elif variant == 'naive':
            iou = naive_iou(boxes[order[:1]], boxes[order[1:]], is_aligned=False, planar=planar).reshape(-1)
        else:
            iou = iou_pairwise(boxes[order[:1]], boxes[order[1:]], variant=variant, planar=planar, nthreads=nthreads).reshape(-1)
        order = order[1:][iou <= np.float32(iou_threshold)]
    return np.asarray(keep, dtype=np.int64)


def batched_nms(boxes, scores, idxs, iou_threshold=0.5, max_num=None, variant='efficient', planar='mmcv'):
    """sph_batched_nms: sphdet/bbox/nms/sph_nms.py:22-60 -> (dets (K', d+1), keep (K',))."""
    boxes = _np(boxes, np.float32)
    scores = _np(scores, np.float32)
    idxs = np.asarray(idxs)
    total = np.zeros(scores.shape, dtype=bool)
    for c in np.unique(idxs):
        m = np.nonzero(idxs == c)[0]
        k = nms_op(boxes[m], scores[m], iou_threshold, variant, planar)
        total[m[k]] = True
    keep = np.nonzero(total)[0]
    inds = np.argsort(-scores[keep], kind='stable')
    keep = keep[inds]
    max_num = boxes.shape[0] if max_num is None else min(max_num, boxes.shape[0])
    keep = keep[:max_num]
    dets = np.concatenate([boxes[keep], scores[keep, None]], axis=-1)
    return dets, keep.astype(np.int64)


def assign_wrt_overlaps(overlaps, gt_labels=None, pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.0,
                        gt_max_assign_all=True, match_low_quality=True):
    """MaxIoUAssigner.assign_wrt_overlaps: mmdet/core/bbox/assigners/max_iou_assigner.py:135-220 (numpy).
    -> (assigned_gt_inds, max_overlaps, argmax_overlaps, gt_max_overlaps, gt_argmax_overlaps, assigned_labels)"""
    ov = np.asarray(overlaps, dtype=np.float32)
    k, n = ov.shape
    gt_inds = np.full(n, -1, dtype=np.int64)
    max_ov, argmax_ov = ov.max(0), ov.argmax(0)           # first maximal index, like torch.max(dim)
    gt_max, gt_argmax = ov.max(1), ov.argmax(1)
    if isinstance(neg_iou_thr, (tuple, list)):
        gt_inds[(max_ov >= neg_iou_thr[0]) & (max_ov < neg_iou_thr[1])] = 0
    else:
        gt_inds[(max_ov >= 0) & (max_ov < np.float32(neg_iou_thr))] = 0
    pos = max_ov >= np.float32(pos_iou_thr)
    gt_inds[pos] = argmax_ov[pos] + 1
    if match_low_quality:
        for i in range(k):
            if gt_max[i] >= np.float32(min_pos_iou):
                if gt_max_assign_all:
                    gt_inds[ov[i] == gt_max[i]] = i + 1
                else:
                    gt_inds[gt_argmax[i]] = i + 1
    labels = None
    if gt_labels is not None:
        labels = np.full(n, -1, dtype=np.int64)
        p = gt_inds > 0
        labels[p] = np.asarray(gt_labels)[gt_inds[p] - 1]
    return gt_inds, max_ov, argmax_ov.astype(np.int64), gt_max, gt_argmax.astype(np.int64), labels


def assign(bboxes, gt_bboxes, iou_fn, gt_bboxes_ignore=None, gt_labels=None, ignore_iof_thr=-1, ignore_wrt_candidates=True,
           **kw):
    """MaxIoUAssigner.assign: mmdet/core/bbox/assigners/max_iou_assigner.py:113-127 (numpy) around a caller-supplied
    `iou_fn(b1, b2, mode)` -> (len(b1), len(b2)); returns (overlaps after the ignore step, assign_wrt_overlaps(...))."""
    ov = np.array(iou_fn(gt_bboxes, bboxes, 'iou'), dtype=np.float32)
    if ignore_iof_thr > 0 and gt_bboxes_ignore is not None and len(gt_bboxes_ignore) and len(bboxes):
        if ignore_wrt_candidates:
            ig = np.asarray(iou_fn(bboxes, gt_bboxes_ignore, 'iof')).max(1)
        else:
            ig = np.asarray(iou_fn(gt_bboxes_ignore, bboxes, 'iof')).max(0)
        ov[:, ig > np.float32(ignore_iof_thr)] = -1
    return ov, assign_wrt_overlaps(ov, gt_labels, **kw)


def generate_boxes(n, seed, box='bfov', alpha=(1, 100), beta=(1, 100), gamma=(-90, 90), theta=(0, 360),
                   phi=(0, 180)):
    """Synthetic boxes of the shape of tests/utils/generate_data.py:31-42 (dtype='float'), numpy RNG."""
    rng = np.random.default_rng(seed)
    u = rng.random((n, 5), dtype=np.float32)
    cols = [u[:, 0] * (theta[1] - theta[0]) + theta[0], u[:, 1] * (phi[1] - phi[0]) + phi[0],
            u[:, 2] * (alpha[1] - alpha[0]) + alpha[0], u[:, 3] * (beta[1] - beta[0]) + beta[0]]
    if box == 'rbfov':
        cols.append(u[:, 4] * (gamma[1] - gamma[0]) + gamma[0])
    return np.stack(cols, axis=1).astype(np.float32)


# ---- box coders: numpy restatement ------------------------------------------------------------------
def coder_encode(proposals, gt, means=None, stds=None, dtype=np.float32):
    """bbox2delta — sphdet/bbox/coder/delta_xywh_sph_bbox_coder.py:116-161 (4 columns) and
    delta_xywha_rsph_bbox_coder.py:116-164 (5 columns: fifth delta = deg2rad(gamma_gt - gamma_proposal))."""
    p = np.asarray(proposals, dtype=dtype)
    g = np.asarray(gt, dtype=dtype)
    dim = p.shape[-1]
    eps = dtype(1e-7)
    pw, ph = np.maximum(p[..., 2], eps), np.maximum(p[..., 3], eps)
    gw, gh = np.maximum(g[..., 2], eps), np.maximum(g[..., 3], eps)
    cols = [(g[..., 0] - p[..., 0]) / pw, (g[..., 1] - p[..., 1]) / ph, np.log(gw / pw), np.log(gh / ph)]
    if dim == 5:
        cols.append((g[..., 4] - p[..., 4]) * dtype(np.pi / 180.0))
    d = np.stack(cols, axis=-1).astype(dtype)
    m = np.zeros(dim, dtype) if means is None else np.asarray(means, dtype)
    s = np.ones(dim, dtype) if stds is None else np.asarray(stds, dtype)
    return ((d - m) / s).astype(dtype)


def coder_decode(rois, deltas, means=None, stds=None, wh_ratio_clip=16 / 1000, clip_border=True, add_ctr_clamp=False,
                 ctr_clamp=32, box_dim=None, dtype=np.float32, grad_boxes=None):
    """delta2bbox — delta_xywh_sph_bbox_coder.py:164-263 / delta_xywha_rsph_bbox_coder.py:167-268.
    With `grad_boxes` also returns J^T grad_boxes w.r.t. `deltas` (torch.clamp passes gradients on [min, max])."""
    r = np.asarray(rois, dtype=dtype)
    dl = np.asarray(deltas, dtype=dtype)
    dim = r.shape[-1] if box_dim is None else box_dim
    n = dl.shape[0]
    nc = dl.shape[1] // dim
    if n == 0:
        return dl if grad_boxes is None else (dl, dl)
    m = np.zeros(dim, dtype) if means is None else np.asarray(means, dtype)
    s = np.ones(dim, dtype) if stds is None else np.asarray(stds, dtype)
    d = dl.reshape(-1, dim) * s + m
    p = np.repeat(r, nc, axis=0)
    eps = dtype(1e-7)
    max_ratio = dtype(abs(np.log(wh_ratio_clip)))
    sxy = p[:, 2:4] * d[:, 0:2]
    dwh = d[:, 2:4]
    g_s = np.ones_like(sxy)
    if add_ctr_clamp:
        c = dtype(ctr_clamp)
        g_s = ((sxy >= -c) & (sxy <= c)).astype(dtype)
        sxy = np.clip(sxy, -c, c)
        g_r = (dwh <= max_ratio).astype(dtype)
        dwh = np.minimum(dwh, max_ratio)
    else:
        g_r = ((dwh >= -max_ratio) & (dwh <= max_ratio)).astype(dtype)
        dwh = np.clip(dwh, -max_ratio, max_ratio)
    xy = p[:, 0:2] + sxy
    wh = p[:, 2:4] * np.exp(dwh)
    cols = [xy[:, 0], xy[:, 1], wh[:, 0], wh[:, 1]]
    if dim == 5:
        cols.append(p[:, 4] + d[:, 4] * dtype(180.0 / np.pi))
    b = np.stack(cols, axis=-1).astype(dtype)
    jac = np.empty_like(b)
    jac[:, 0:2] = g_s * p[:, 2:4] * s[0:2]
    jac[:, 2:4] = g_r * wh * s[2:4]
    if dim == 5:
        jac[:, 4] = dtype(180.0 / np.pi) * s[4]
    if clip_border:
        lo = np.array([eps, eps, eps, eps, dtype(-90.0 + 1e-7)][:dim], dtype)
        hi = np.array([dtype(360.0 - 1e-7), dtype(180.0 - 1e-7), dtype(180.0 - 1e-7), dtype(180.0 - 1e-7),
                       dtype(90.0 - 1e-7)][:dim], dtype)
        jac = jac * ((b >= lo) & (b <= hi))
        b = np.clip(b, lo, hi)
    b = b.reshape(n, -1)
    if grad_boxes is None:
        return b
    return b, (np.asarray(grad_boxes, dtype).reshape(-1, dim) * jac).reshape(n, -1).astype(dtype)


# ---- Unbiased IoU (the default backend of SphOverlaps2D and an SphNMS calculator) ---------------------
UNBIASED_PREC = {'f64': 0, 'kernel': 1, 'reference_f32': 2}


def unbiased_iou(b1, b2, is_aligned=True, prec='kernel', nthreads=None):
    """unbiased_iou — sphdet/iou/sph_iou_api.py:103-126 over unbiased_iou_bfov.py / unbiased_iou_rbfov.py.
    prec: 'f64' (reference on float64 tensors), 'kernel' (fp32 jitter + deg2rad, then double: what the HIP kernel
    computes), 'reference_f32' (the reference's mixed numpy arithmetic on float32 tensors)."""
    a = _np(b1, np.float64)
    b = _np(b2, np.float64)
    dim = a.shape[1]
    if not is_aligned:
        m, n = a.shape[0], b.shape[0]
        a, b = np.repeat(a, n, axis=0), np.tile(b, (m, 1))
    out = np.empty(a.shape[0], np.float64)
    fn = lib().sph2pob_oracle_unbiased_iou
    fn.argtypes = [ctypes.POINTER(ctypes.c_double)] * 3 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    rc = fn(_ptr(np.ascontiguousarray(a), ctypes.c_double), _ptr(np.ascontiguousarray(b), ctypes.c_double),
            _ptr(out, ctypes.c_double), a.shape[0], dim, UNBIASED_PREC[prec], nthreads or max_threads())
    assert rc == 0, rc
    out = out.astype(np.float32 if prec != 'f64' else np.float64)
    return out if is_aligned else out.reshape(m, n)


def naive_iou(b1, b2, is_aligned=True, planar='mmcv', box_formator='sph2pix'):
    """naive_iou — sphdet/iou/sph_iou_api.py:179-197: Sph2PlanarBoxTransform('sph2pix'), img_size (512, 1024)
    (sphdet/bbox/box_formator.py:76-83, :161-178) then mmcv.ops.bbox_overlaps (BFoV, xyxy, offset 0) or
    mmcv.ops.box_iou_rotated (RBFoV, angle = -deg2rad(gamma)).  mmcv-full 1.6.0 is absent: bbox_overlaps is restated
    from its published kernel (inter / max(a1 + a2 - inter, offset)) — PARITY UNPINNED for the BFoV branch; the RBFoV
    branch goes through the same planar restatement as the Sph2Pob path (`planar`)."""
    a = _np(b1, np.float32)
    b = _np(b2, np.float32)
    dim = a.shape[1]
    m, n = a.shape[0], b.shape[0]
    if not is_aligned:
        a, b = np.repeat(a, n, axis=0), np.tile(b, (m, 1))
    f = np.float32

    def pix(x):
        if box_formator == 'sph2tan':   # box_formator.py:98-106: w = 2R tan(alpha / 2), 2R = img_w / pi
            two_r = f(1024 / np.pi)
            w = two_r * np.tan((x[:, 2] * f(np.pi / 180)) / f(2)).astype(f)
            h = two_r * np.tan((x[:, 3] * f(np.pi / 180)) / f(2)).astype(f)
        else:
            assert box_formator == 'sph2pix'
            w, h = (x[:, 2] / f(360)) * f(1024), (x[:, 3] / f(180)) * f(512)
        return np.stack([(x[:, 0] / f(360)) * f(1024), (x[:, 1] / f(180)) * f(512), w, h], axis=1).astype(f)
    pa, pb = pix(a), pix(b)
    if dim == 4:
        def xyxy(p):
            return p[:, 0] - p[:, 2] / f(2), p[:, 1] - p[:, 3] / f(2), p[:, 0] + p[:, 2] / f(2), p[:, 1] + p[:, 3] / f(2)
        ax1, ay1, ax2, ay2 = xyxy(pa)
        bx1, by1, bx2, by2 = xyxy(pb)
        iw = np.maximum(np.minimum(ax2, bx2) - np.maximum(ax1, bx1), f(0))
        ih = np.maximum(np.minimum(ay2, by2) - np.maximum(ay1, by1), f(0))
        inter = iw * ih
        with np.errstate(divide='ignore', invalid='ignore'):
            out = inter / np.maximum((ax2 - ax1) * (ay2 - ay1) + (bx2 - bx1) * (by2 - by1) - inter, f(0))
    else:
        ra = np.concatenate([pa, -(a[:, 4:5] * f(np.pi / 180))], axis=1).astype(f)
        rb = np.concatenate([pb, -(b[:, 4:5] * f(np.pi / 180))], axis=1).astype(f)
        out = planar_iou(ra, rb, mode='iou', planar=planar)
    return out if is_aligned else out.reshape(m, n)


# ---- Sph2PobL1Loss -----------------------------------------------------------------------------------
def obb_l1_elements(pred, target, encode=True, swap=False, angle_modifier='original', dtype=np.float32):
    """Unweighted (n, 5) element losses of Sph2PobL1Loss — sphdet/losses/sph2pob_l1_loss.py:28-88 behind
    Sph2PobTransfrom (sph2pob_transform.py:24-35): |bbox2delta(planar pred, planar target)| (or the swapped roles),
    |planar pred - planar target| when encode=False."""
    p, t = transform(pred, target, variant='standard', jitter=True, dtype=dtype)
    p, t = p.astype(dtype), t.astype(dtype)
    if not encode:
        return np.abs(p - t)
    pr, gt = (t, p) if swap else (p, t)
    eps = dtype(1e-7)
    pw, ph = np.maximum(pr[:, 2], eps), np.maximum(pr[:, 3], eps)
    gw, gh = np.maximum(gt[:, 2], eps), np.maximum(gt[:, 3], eps)
    pi = dtype(np.pi)

    def wrap(a):
        return a if angle_modifier == 'original' else np.mod(a + pi, pi)
    d = np.stack([(gt[:, 0] - pr[:, 0]) / pw, (gt[:, 1] - pr[:, 1]) / ph, np.log(gw / pw), np.log(gh / ph),
                  (wrap(gt[:, 4]) - wrap(pr[:, 4])) / pi], axis=-1)
    return np.abs(d).astype(dtype)

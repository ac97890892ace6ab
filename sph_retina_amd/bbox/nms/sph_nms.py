"""Spherical NMS — mirrors sphdet/bbox/nms/sph_nms.py:7-74 (SphNMS, sph_batched_nms, sph_nms_op).

The reference loops in Python: per class, per kept box, one full Sph2Pob IoU pipeline (1 x K) and a host sync.
Here a call of up to 16 384 candidates is ONE launcher call and four kernels with no host work (`sph2pob_batched_nms_f32`:
composite-key sort in LDS, suppression bit-matrix with the Sph2Pob IoU evaluated once per same-class pair, per-class greedy
sweeps, score-ordered selection) and one host read — the number of kept boxes, which sizes the result.  Larger calls sort
with torch and run the same two NMS kernels; a class of more than 32 704 boxes is swept in chunks.

Kept behaviour: accepted calculator names, `iou <= thr` keeps, per-class suppression regardless of
`class_agnostic` (the reference pops the flag but never uses it, :34,44), final ordering by descending score,
`max_num`, dets = cat(boxes, scores).  Differences: ties are broken deterministically (stable sorts; the reference's
`torch.argsort(descending=True)` is unstable); `nms_cfg=None` raises instead of `sys.exit()` (:24-28).
"""
import ctypes

import torch

from ... import _lib
from ... import _torch_glue as G

_CALCULATORS = {'sph2pob_efficient': 'efficient', 'sph2pob_efficient_iou': 'efficient',
                'sph2pob_standard': 'standard', 'sph2pob_standard_iou': 'standard',
                'unbiased_iou': 'unbiased', 'naive_iou': 'naive'}  # sph_nms.py:9-14


def _variant_of(iou_calculator):
    """Calculator spelling -> kernel variant.  The reference passes either the registry string (SphNMS, sph_nms.py:8-16)
    or the IoU FUNCTION itself (`sph_nms_op(boxes, scores, thr, sph2pob_efficient_iou)`, :19, :62): both are accepted;
    a function is mapped by its name."""
    if isinstance(iou_calculator, str):
        name = iou_calculator
    elif callable(iou_calculator):
        name = getattr(iou_calculator, '__name__', '')
    else:
        raise TypeError(f'iou_calculator must be a name or one of the sph IoU functions, got {type(iou_calculator).__name__}')
    if name in _CALCULATORS:
        return _CALCULATORS[name]
    if name in _CALCULATORS.values() or name == 'naive_tan':   # already a variant name ('efficient', ...): SphNMS hands these over
        return name
    raise TypeError(f'Not supported iou_calculator: {name!r} (accepted: {sorted(_CALCULATORS)})')


def _nms_sorted(boxes_sorted, cls_sorted, iou_threshold, variant):
    """keep flags (uint8, K) for boxes sorted by (class, -score)."""
    k, dim = boxes_sorted.shape
    lib = _lib.lib()
    dev = boxes_sorted.device
    keep = torch.empty((k,), dtype=torch.uint8, device=dev)
    if k == 0:
        return keep
    # the suppression matrix is k x (largest class segment / 64 + 2) words: its width needs the largest segment on the
    # host (one sync; the caller's nonzero() syncs anyway).  multiclass_nms hands over every (box, class) candidate
    # above score_thr — far more than 32 768 rows in total, but a class segment stays small.
    # Small inputs take the full k x k/64 layout (<= 8 MB) and skip that sync.
    max_seg = k if cls_sorted is None or k <= 8192 or dev.type == 'cpu' else int(torch.unique_consecutive(cls_sorted, return_counts=True)[1].max())
    limit = lib.sph2pob_nms_max_boxes()
    if dev.type == 'cpu':   # the host twin keeps no suppression matrix: no workspace, no per-class limit
        G.call('sph2pob_nms_segmented_f32', dev, G.ptr(boxes_sorted), G.ptr(cls_sorted), ctypes.c_int64(k), dim, G.VARIANTS[variant],
               ctypes.c_float(iou_threshold), ctypes.c_int64(0), None, G.ptr(keep), None)
        return keep
    if max_seg > limit:
        # the sweep keeps one class segment's "removed" bit-vector in LDS: 32 704 boxes per class; the
        # reference has no limit but needs one host round trip per kept box (minutes at this size)
        raise ValueError(f'the sweep kernel takes at most {limit} boxes per class segment, got {max_seg}: callers route longer '
                         'classes through _nms_one_class_chunked')
    ws = torch.empty((lib.sph2pob_nms_segmented_workspace_bytes(k, max_seg) // 8,), dtype=torch.int64, device=dev)
    G.call('sph2pob_nms_segmented_f32', dev, G.ptr(boxes_sorted), G.ptr(cls_sorted), ctypes.c_int64(k), dim,
           G.VARIANTS[variant], ctypes.c_float(iou_threshold), ctypes.c_int64(max_seg), G.ptr(ws), G.ptr(keep),
           G.stream_of(boxes_sorted))
    return keep


def _fused_nms(boxes, scores, idxs, iou_threshold, max_num, variant):
    """The host-free route (k <= 16 384) -> (dets, keep), or None when a class id does not fit the composite sort key."""
    lib = _lib.lib()
    k, dim = boxes.shape
    dev = boxes.device
    b, sc = G.as_f32_nograd(boxes), G.as_f32_nograd(scores)
    ix = None if idxs is None else idxs.to(torch.int64).contiguous()
    rows = max(min(int(max_num), k), 0)
    need = lib.sph2pob_batched_nms_workspace_bytes(k, dim)
    ws = G.scratch(dev, need + 256)                      # cached per (device, stream); its tail holds the status word
    status = ws[need:need + 4].view(torch.int32)
    keep = torch.empty((rows,), dtype=torch.int64, device=dev)
    dets = torch.empty((rows, dim + 1), dtype=torch.float32, device=dev)
    G.call('sph2pob_batched_nms_f32', dev, G.ptr(b), G.ptr(sc), G.ptr(ix), k, dim, G.VARIANTS[variant], float(iou_threshold), rows,
           G.ptr(ws), G.ptr(keep), G.ptr(dets), G.ptr(status), G.raw_stream_of(dev))
    n = int(status.item())   # the one host read: the result's length
    if n < 0:
        return None
    dets = dets[:n]
    if boxes.dtype != torch.float32 and boxes.is_floating_point():
        dets = dets.to(boxes.dtype)
    return dets, keep[:n]


def _pairwise_fn(variant):
    from ...iou import sph_iou_api as A
    return {'efficient': A.sph2pob_efficient_iou, 'standard': A.sph2pob_standard_iou, 'unbiased': A.unbiased_iou,
            'naive': A.naive_iou, 'naive_tan': lambda a, b: A.naive_iou(a, b, box_formator='sph2tan')}[variant]


def _nms_one_class_chunked(boxes_sorted, iou_threshold, variant, chunk=16384):
    """Greedy NMS of ONE class of any length, boxes in descending-score order -> keep flags (bool).  Chunks of `chunk`
    boxes: a chunk's boxes are first tested against every box kept so far (one pairwise IoU launch, rows = the kept boxes
    in the bboxes1 role, as sph_nms_op has them, sph_nms.py:70), the survivors go through the NMS kernels.  The reference
    has no size limit (one IoU pipeline + host sync per kept box: minutes at this size); the sweep kernel's limit is 32 704."""
    n = boxes_sorted.size(0)
    flags = torch.zeros((n,), dtype=torch.bool, device=boxes_sorted.device)
    kept = None
    iou = _pairwise_fn(variant)
    for lo in range(0, n, chunk):
        part = boxes_sorted[lo:lo + chunk]
        alive = None
        if kept is not None and kept.size(0) > 0:
            alive = (iou(kept, part) <= iou_threshold).all(dim=0)   # `iou <= thr` keeps (a NaN IoU suppresses, as in the reference)
            part = part[alive]
        if part.size(0) == 0:
            continue
        f = _nms_sorted(part.contiguous(), None, iou_threshold, variant).bool()
        pos = torch.arange(lo, min(lo + chunk, n), device=flags.device)
        if alive is not None:
            pos = pos[alive]
        flags[pos[f]] = True
        kept = part[f] if kept is None else torch.cat([kept, part[f]])
    return flags


def sph_nms_op(boxes, scores, iou_threshold, iou_calculator='sph2pob_efficient'):
    """Single-class greedy NMS -> indices of kept boxes in descending-score order (reference :62-74)."""
    variant = _variant_of(iou_calculator)
    assert boxes.size(1) in [4, 5]
    G.require_hip(boxes, scores)
    k = boxes.size(0)
    if boxes.is_cuda and 0 < k <= _lib.lib().sph2pob_batched_nms_max_boxes():
        return _fused_nms(boxes, scores, None, iou_threshold, k, variant)[1]
    order = torch.argsort(scores, descending=True, stable=True)
    bs = G.as_f32(boxes[order])
    if boxes.is_cuda and k > _lib.lib().sph2pob_nms_max_boxes():
        flags = _nms_one_class_chunked(bs, float(iou_threshold), variant)
    else:
        flags = _nms_sorted(bs, None, float(iou_threshold), variant).bool()
    return order[flags]


def sph_batched_nms(boxes, scores, idxs, nms_cfg, iou_calculator='efficient', class_agnostic=False):
    """Reference sph_batched_nms (:22-60) -> (dets (K', d+1), keep (K',) int64 indices into the input)."""
    if nms_cfg is None:
        raise ValueError('nms_cfg is None (the reference prints a message and calls sys.exit() here)')
    nms_cfg_ = nms_cfg.copy()
    nms_cfg_.pop('class_agnostic', class_agnostic)  # accepted and ignored, as in the reference
    nms_cfg_.pop('type', 'nms')
    nms_cfg_.pop('split_thr', 10000)
    iou_threshold = nms_cfg_.pop('iou_threshold', 0.5)
    max_num = min(nms_cfg_.pop('max_num', boxes.shape[0]), boxes.shape[0])
    G.require_hip(boxes, scores, idxs)
    assert boxes.size(1) in [4, 5]
    variant = _variant_of(iou_calculator)
    k = boxes.size(0)
    lib = _lib.lib()
    if boxes.is_cuda and 0 < k <= lib.sph2pob_batched_nms_max_boxes() and not idxs.is_floating_point():
        out = _fused_nms(boxes, scores, idxs, iou_threshold, max_num, variant)
        if out is not None:
            return out
    # the general route: sort by (class ascending, score descending) with two stable torch sorts
    by_score = torch.argsort(scores, descending=True, stable=True)
    order = by_score[torch.argsort(idxs[by_score], stable=True)]
    cls_sorted = idxs[order].to(torch.int64).contiguous()
    boxes_sorted = G.as_f32(boxes[order])
    limit = lib.sph2pob_nms_max_boxes()
    counts = torch.unique_consecutive(cls_sorted, return_counts=True)[1] if k > 8192 and boxes.is_cuda else None
    if counts is not None and int(counts.max()) > limit:
        # a class longer than the sweep's limit: that class in chunks, the others (as one call) through the kernels
        bounds = torch.cumsum(counts, 0).tolist()
        flags = torch.zeros((k,), dtype=torch.uint8, device=boxes.device)
        small = torch.ones((k,), dtype=torch.bool, device=boxes.device)
        lo = 0
        for hi in bounds:
            if hi - lo > limit:
                flags[lo:hi] = _nms_one_class_chunked(boxes_sorted[lo:hi], float(iou_threshold), variant).to(torch.uint8)
                small[lo:hi] = False
            lo = hi
        if bool(small.any()):
            flags[small] = _nms_sorted(boxes_sorted[small].contiguous(), cls_sorted[small].contiguous(), float(iou_threshold), variant)
    else:
        flags = _nms_sorted(boxes_sorted, cls_sorted, float(iou_threshold), variant)
    # the reference takes the kept original indices in ascending order (:49) and sorts them by descending score (:50-51);
    # a stable sort breaks ties by that ascending index — which is the order `by_score` already has, so the kept entries
    # of `by_score` ARE that list: one scatter + one masked gather instead of nonzero + gather + a third sort
    total_mask = torch.empty(scores.shape, dtype=torch.bool, device=scores.device)
    total_mask[order] = flags.bool()
    keep = by_score[total_mask[by_score]][:max_num]
    dets = torch.cat([boxes[keep], scores[keep][:, None]], -1)
    return dets, keep


class SphNMS:
    """SphNMS(iou_calculator='sph2pob_efficient')(boxes, scores, idxs, nms_cfg, class_agnostic=False)."""

    def __init__(self, iou_calculator='sph2pob_efficient'):
        if iou_calculator in _CALCULATORS:
            self.variant = _CALCULATORS[iou_calculator]
        else:
            raise TypeError('Not supported iou_calculator.')  # the reference's `raise NotImplemented(...)`
        self.iou_calculator = iou_calculator

    def __call__(self, boxes, scores, idxs, nms_cfg, class_agnostic=False):
        return sph_batched_nms(boxes, scores, idxs, nms_cfg, self.variant, class_agnostic)


class PlanarNMS:
    """PlanarNMS(box_formator='sph2pix')(boxes, scores, idxs, nms_cfg, class_agnostic=True) — reference
    sphdet/bbox/nms/planar_nms.py:7-19: the boxes drawn in ERP pixels (Sph2PlanarBoxTransform) go through mmcv's
    `batched_nms`, i.e. greedy NMS on the planar IoU — the Naive-IoU of this package — and, unlike SphNMS, across classes
    by default (`class_agnostic=True`, overridable from `nms_cfg` as in mmcv).  Served by the same two kernels with the
    naive variant.  mmcv is absent here: `batched_nms` is restated from its published behaviour (suppress IoU > thr,
    result in descending score order, `max_num`) — parity unpinned."""

    def __init__(self, box_formator='sph2pix'):
        assert box_formator in ['sph2pix', 'sph2tan']   # Sph2PlanarBoxTransform.__init__ (box_formator.py:163)
        self.box_formator = box_formator

    def __call__(self, boxes, scores, idxs, nms_cfg, class_agnostic=True):
        nms_cfg_ = dict(nms_cfg)
        class_agnostic = nms_cfg_.pop('class_agnostic', class_agnostic)
        if class_agnostic:
            idxs = torch.zeros_like(idxs)
        return sph_batched_nms(boxes, scores, idxs, nms_cfg_, 'naive' if self.box_formator == 'sph2pix' else 'naive_tan', class_agnostic)

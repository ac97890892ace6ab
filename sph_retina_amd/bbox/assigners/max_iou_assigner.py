"""MaxIoUAssigner with a fused MI355X epilogue (the immediate consumer of the pairwise IoU kernel).

`SphMaxIoUAssigner` mirrors mmdet's `MaxIoUAssigner` (mmdet/core/bbox/assigners/max_iou_assigner.py:11-220): same
constructor arguments, `assign(bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None)` and
`assign_wrt_overlaps(overlaps, gt_labels=None)`.  The overlaps come from the Sph2Pob pairwise kernel
(`iou_calculator(gt_bboxes, bboxes)`, :113); everything after it — both `max` reductions, the threshold steps and the
python `for i in range(num_gts)` low-quality loop, which costs one device->host sync per GT in the reference
(:200-207) — is three kernel launches (`sph2pob_assign_f32`) with no host synchronisation.

With a closed-form Sph2Pob calculator (`sph2pob_standard_iou` / `sph2pob_efficient_iou`, default arithmetic) `assign` goes
one step further and never materialises the (k, n) overlaps (`sph2pob_iou_assign_f32`): the pairwise kernel keeps the
per-anchor and per-GT maxima while it finishes the pairs, and the low-quality step re-evaluates a GT row only against the
column tile that holds its maximum.  Bit-identical to the matrix route (tests/test_gpu_assigner.py).
`gpu_assign_thr` (:100-110, :128-133) moves an assignment with more GTs than the threshold to the CPU, as the reference does —
served there by the product's host twins (libsph2pob_host.so) — and returns the result on the boxes' device.
"""
import ctypes

import torch

from ... import _lib
from ... import _torch_glue as G
from ...registry import BBOX_ASSIGNERS, build_iou_calculator


class _AssignResult:
    """Stand-in for mmdet's AssignResult (mmdet/core/bbox/assigners/assign_result.py) when mmdet is not importable:
    the fields and the methods samplers use (`add_gt_`, extra properties, `info`)."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts = num_gts
        self.gt_inds = gt_inds
        self.max_overlaps = max_overlaps
        self.labels = labels
        self._extra_properties = {}

    @property
    def num_preds(self):
        return len(self.gt_inds)

    def set_extra_property(self, key, value):
        assert key not in self.info
        self._extra_properties[key] = value

    def get_extra_property(self, key):
        return self._extra_properties.get(key, None)

    @property
    def info(self):
        basic = {'num_gts': self.num_gts, 'num_preds': self.num_preds, 'gt_inds': self.gt_inds,
                 'max_overlaps': self.max_overlaps, 'labels': self.labels}
        basic.update(self._extra_properties)
        return basic

    def add_gt_(self, gt_labels):  # assign_result.py:192-206
        self_inds = torch.arange(1, len(gt_labels) + 1, dtype=torch.long, device=gt_labels.device)
        self.gt_inds = torch.cat([self_inds, self.gt_inds])
        self.max_overlaps = torch.cat([self.max_overlaps.new_ones(len(gt_labels)), self.max_overlaps])
        if self.labels is not None:
            self.labels = torch.cat([gt_labels, self.labels])


try:  # hand mmdet's samplers their own class when mmdet is present
    from mmdet.core.bbox.assigners.assign_result import AssignResult
except Exception:  # mmdet / mmcv not installed (this container)
    AssignResult = _AssignResult


def assign_wrt_overlaps(overlaps, gt_labels=None, pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.0,
                        gt_max_assign_all=True, match_low_quality=True, return_extras=False):
    """assign_wrt_overlaps (max_iou_assigner.py:135-220) on a (k, n) overlaps matrix living on the MI355X."""
    num_gts, num_bboxes = overlaps.size(0), overlaps.size(1)
    if num_gts == 0 or num_bboxes == 0:  # :148-165
        gt_inds = overlaps.new_full((num_bboxes,), -1, dtype=torch.long)
        if num_gts == 0:
            gt_inds[:] = 0
        labels = None if gt_labels is None else overlaps.new_full((num_bboxes,), -1, dtype=torch.long)
        return AssignResult(num_gts, gt_inds, overlaps.new_zeros((num_bboxes,)), labels=labels)
    G.require_hip(overlaps)
    ov = G.as_f32(overlaps.detach())
    dev = ov.device
    neg_lo, neg_hi = _thresholds(neg_iou_thr)
    max_ov = torch.empty((num_bboxes,), dtype=torch.float32, device=dev)
    argmax_ov = torch.empty((num_bboxes,), dtype=torch.int64, device=dev)
    gt_max = torch.empty((num_gts,), dtype=torch.float32, device=dev)
    gt_argmax = torch.empty((num_gts,), dtype=torch.int64, device=dev)
    gt_inds = torch.empty((num_bboxes,), dtype=torch.int64, device=dev)
    labels = gl = None
    if gt_labels is not None:
        gl = gt_labels.to(device=dev, dtype=torch.int64).contiguous()
        labels = torch.empty((num_bboxes,), dtype=torch.int64, device=dev)
    ws = torch.empty((_lib.lib().sph2pob_assign_workspace_bytes(num_gts, num_bboxes) // 8 if dev.type != 'cpu' else 1,), dtype=torch.int64, device=dev)
    G.call('sph2pob_assign_f32', dev, G.ptr(ov), ctypes.c_int64(num_gts), ctypes.c_int64(num_bboxes),
           ctypes.c_float(pos_iou_thr), ctypes.c_float(neg_lo), ctypes.c_float(neg_hi), ctypes.c_float(min_pos_iou),
           int(bool(match_low_quality)), int(bool(gt_max_assign_all)), G.ptr(gl), G.ptr(max_ov), G.ptr(argmax_ov),
           G.ptr(gt_max), G.ptr(gt_argmax), G.ptr(gt_inds), G.ptr(labels), G.ptr(ws), G.stream_of(ov))
    res = AssignResult(num_gts, gt_inds, max_ov if overlaps.dtype == torch.float32 else max_ov.to(overlaps.dtype), labels)
    if return_extras:
        return res, dict(argmax_overlaps=argmax_ov, gt_max_overlaps=gt_max, gt_argmax_overlaps=gt_argmax)
    return res


def _thresholds(neg_iou_thr):
    if isinstance(neg_iou_thr, (tuple, list)):
        assert len(neg_iou_thr) == 2
        return float(neg_iou_thr[0]), float(neg_iou_thr[1])
    return 0.0, float(neg_iou_thr)


_FUSED_BACKENDS = {'sph2pob_standard_iou': 'standard', 'sph2pob_efficient_iou': 'efficient'}


def fused_assign(gt_bboxes, bboxes, gt_labels=None, variant='standard', pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.0,
                 gt_max_assign_all=True, match_low_quality=True, ignore_mask=None, rbb_edge='arc', return_overlaps=False,
                 return_extras=False):
    """`iou_calculator(gt_bboxes, bboxes)` + `assign_wrt_overlaps` (max_iou_assigner.py:113, :135-220) in three launches
    without the (k, n) matrix (`return_overlaps=True` also writes it).  `ignore_mask`: optional (n,) bool, the columns the
    reference sets to -1 (:115-126).  k > 0 and n > 0; boxes (k, 4|5) / (n, 4|5) on the MI355X."""
    G.require_hip(gt_bboxes, bboxes)
    if not bboxes.is_cuda:
        raise RuntimeError('fused_assign runs on MI355X tensors (on the CPU: iou_calculator + assign_wrt_overlaps)')
    gt, bx = G.as_f32_nograd(gt_bboxes), G.as_f32_nograd(bboxes)
    k, n, dim = gt.size(0), bx.size(0), gt.size(1)
    assert bx.size(1) == dim and dim in (4, 5) and k > 0 and n > 0
    dev = bx.device
    neg_lo, neg_hi = _thresholds(neg_iou_thr)
    max_ov = torch.empty((n,), dtype=torch.float32, device=dev)
    gt_inds = torch.empty((n,), dtype=torch.int64, device=dev)
    labels = gl = argmax_ov = gt_max = gt_argmax = ov = ign = None
    if gt_labels is not None:
        gl = gt_labels.to(device=dev, dtype=torch.int64).contiguous()
        labels = torch.empty((n,), dtype=torch.int64, device=dev)
    if return_extras:
        argmax_ov = torch.empty((n,), dtype=torch.int64, device=dev)
        gt_max = torch.empty((k,), dtype=torch.float32, device=dev)
        gt_argmax = torch.empty((k,), dtype=torch.int64, device=dev)
    if return_overlaps:
        ov = torch.empty((k, n), dtype=torch.float32, device=dev)
    if ignore_mask is not None:
        ign = ignore_mask.to(device=dev, dtype=torch.uint8).contiguous()
    lib = _lib.lib()
    ws, state = G.assign_workspace(dev, lib.sph2pob_iou_assign_workspace_bytes(k, n), lib.sph2pob_iou_assign_state_bytes(k, n))
    try:
        G.call('sph2pob_iou_assign_f32', dev, G.ptr(gt), k, G.ptr(bx), n, dim, G.VARIANTS[variant], G.EDGES[rbb_edge], G.ptr(ign),
               G.ptr(ov), pos_iou_thr, neg_lo, neg_hi, min_pos_iou, int(bool(match_low_quality)), int(bool(gt_max_assign_all)),
               G.ptr(gl), G.ptr(max_ov), G.ptr(argmax_ov), G.ptr(gt_max), G.ptr(gt_argmax), G.ptr(gt_inds), G.ptr(labels),
               G.ptr(ws), G.ptr(state), G.raw_stream_of(dev))
    except Exception:
        G.drop_assign_workspace(dev)
        raise
    res = AssignResult(k, gt_inds, max_ov if bboxes.dtype == torch.float32 or not bboxes.is_floating_point()
                       else max_ov.to(bboxes.dtype), labels)
    out = (res,)
    if return_overlaps:
        out += (ov,)
    if return_extras:
        out += (dict(argmax_overlaps=argmax_ov, gt_max_overlaps=gt_max, gt_argmax_overlaps=gt_argmax),)
    return out[0] if len(out) == 1 else out


@BBOX_ASSIGNERS.register_module(force=True)
class SphMaxIoUAssigner:
    """Same constructor as mmdet's MaxIoUAssigner (:45-65); `iou_calculator` defaults to the Sph2Pob standard IoU."""

    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True, ignore_iof_thr=-1,
                 ignore_wrt_candidates=True, match_low_quality=True, gpu_assign_thr=-1,
                 iou_calculator=dict(type='SphOverlaps2D', backend='sph2pob_standard_iou', box_version=4), fused=True):
        self.fused = fused   # False: always go through the (k, n) matrix (A/B and tests)
        self.pos_iou_thr = pos_iou_thr
        self.neg_iou_thr = neg_iou_thr
        self.min_pos_iou = min_pos_iou
        self.gt_max_assign_all = gt_max_assign_all
        self.ignore_iof_thr = ignore_iof_thr
        self.ignore_wrt_candidates = ignore_wrt_candidates
        self.gpu_assign_thr = gpu_assign_thr
        self.match_low_quality = match_low_quality
        self.iou_calculator = build_iou_calculator(iou_calculator) if isinstance(iou_calculator, dict) else iou_calculator

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        if self.gpu_assign_thr > 0 and gt_bboxes.shape[0] > self.gpu_assign_thr and bboxes.is_cuda:   # :100-110, :128-133
            device = bboxes.device
            res = self.assign(bboxes.cpu(), gt_bboxes.cpu(), None if gt_bboxes_ignore is None else gt_bboxes_ignore.cpu(),
                              None if gt_labels is None else gt_labels.cpu())
            res.gt_inds, res.max_overlaps = res.gt_inds.to(device), res.max_overlaps.to(device)
            if res.labels is not None:
                res.labels = res.labels.to(device)
            return res
        ignore_mask = None
        if (self.ignore_iof_thr > 0 and gt_bboxes_ignore is not None and gt_bboxes_ignore.numel() > 0
                and bboxes.numel() > 0):  # :115-126
            if self.ignore_wrt_candidates:
                ignore_max, _ = self.iou_calculator(bboxes, gt_bboxes_ignore, mode='iof').max(dim=1)
            else:
                ignore_max, _ = self.iou_calculator(gt_bboxes_ignore, bboxes, mode='iof').max(dim=0)
            ignore_mask = ignore_max > self.ignore_iof_thr
        variant = self._fused_variant(bboxes, gt_bboxes)
        if variant is not None:   # no (k, n) matrix
            v = self.iou_calculator.box_version
            return fused_assign(gt_bboxes[..., :v], bboxes[..., :v], gt_labels, variant, self.pos_iou_thr, self.neg_iou_thr,
                                self.min_pos_iou, self.gt_max_assign_all, self.match_low_quality, ignore_mask)
        overlaps = self.iou_calculator(gt_bboxes, bboxes)  # rows = GT (:113)
        if ignore_mask is not None:
            overlaps[:, ignore_mask] = -1
        return self.assign_wrt_overlaps(overlaps, gt_labels)

    def _fused_variant(self, bboxes, gt_bboxes):
        """The closed-form variant the fused path serves, or None (another calculator / backend / arithmetic, empty inputs)."""
        from ...iou.sph_iou_calculator import SphOverlaps2D
        c = self.iou_calculator
        if not self.fused or type(c) is not SphOverlaps2D or c.backend not in _FUSED_BACKENDS or G.get_arithmetic() == 'reference':
            return None
        if gt_bboxes.size(0) == 0 or bboxes.size(0) == 0 or c.box_version not in (4, 5) or not bboxes.is_cuda:
            return None
        if bboxes.size(-1) < c.box_version or gt_bboxes.size(-1) < c.box_version or gt_bboxes.size(0) > 262140:
            return None
        return _FUSED_BACKENDS[c.backend]

    def assign_wrt_overlaps(self, overlaps, gt_labels=None):
        return assign_wrt_overlaps(overlaps, gt_labels, self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou,
                                   self.gt_max_assign_all, self.match_low_quality)

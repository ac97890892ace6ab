"""Spherical delta box coders on the MI355X — the step in front of `loss_bbox` when
`reg_decoded_bbox=True` (sphdet/models/heads/sph_retina_head.py:255-264).

Mirrors `DeltaXYWHSphBBoxCoder` (sphdet/bbox/coder/delta_xywh_sph_bbox_coder.py:10-113) and `DeltaXYWHASphBBoxCoder`
(sphdet/bbox/coder/delta_xywha_rsph_bbox_coder.py:10-113): same constructor arguments, `encode(bboxes, gt_bboxes)`,
`decode(bboxes, pred_bboxes, max_shape=None, wh_ratio_clip=16/1000)`, and the module-level `bbox2delta` /
`delta2bbox` (:116-161, :164-263).  One kernel each (`sph2pob_coder_encode_f32`, `sph2pob_coder_decode_f32`); decode is
a `torch.autograd.Function` whose backward is `sph2pob_coder_decode_bwd_f32`, so gradients of a loss on decoded boxes
reach the network's deltas exactly as the reference's autograd graph delivers them (clamp gates included).
`encode` produces regression targets and is not differentiable here (the reference never back-propagates through it).
"""
import ctypes
import math

import torch

from ... import _torch_glue as G
from ...registry import BBOX_CODERS


def _host_vec(values, dim):
    values = tuple(float(v) for v in values)
    assert len(values) == dim, f'expected {dim} means/stds, got {len(values)}'
    return (ctypes.c_float * dim)(*values)


def bbox2delta(proposals, gt, means=None, stds=None):
    """Deltas that move `proposals` onto `gt` (delta_xywh_sph_bbox_coder.py:116-161 / rsph :116-164)."""
    assert proposals.size() == gt.size()
    dim = proposals.size(-1)
    assert dim in (4, 5)
    means = (0.,) * dim if means is None else means
    stds = (1.,) * dim if stds is None else stds
    G.require_hip(proposals, gt)
    p = G.as_f32(proposals.detach()).reshape(-1, dim)
    g = G.as_f32(gt.detach()).reshape(-1, dim)
    out = torch.empty_like(p)
    n = p.size(0)
    if n:
        G.call('sph2pob_coder_encode_f32', p.device, G.ptr(p), G.ptr(g), _host_vec(means, dim), _host_vec(stds, dim),
               G.ptr(out), n, dim, G.stream_of(p))
    return out.reshape(proposals.shape)


class _DecodeFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rois, deltas, means, stds, num_classes, dim, max_ratio, flags, ctr_clamp):
        n = rois.size(0)
        out = torch.empty_like(deltas)
        G.call('sph2pob_coder_decode_f32', rois.device, G.ptr(rois), G.ptr(deltas), _host_vec(means, dim),
               _host_vec(stds, dim), G.ptr(out), n, num_classes, dim, max_ratio, flags, ctr_clamp, G.stream_of(rois))
        ctx.save_for_backward(rois, deltas)
        ctx.cfg = (means, stds, num_classes, dim, max_ratio, flags, ctr_clamp)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        rois, deltas = ctx.saved_tensors
        means, stds, num_classes, dim, max_ratio, flags, ctr_clamp = ctx.cfg
        g = G.as_f32(grad_out)
        gd = torch.empty_like(deltas)
        G.call('sph2pob_coder_decode_bwd_f32', rois.device, G.ptr(rois), G.ptr(deltas), G.ptr(g), _host_vec(means, dim),
               _host_vec(stds, dim), G.ptr(gd), rois.size(0), num_classes, dim, max_ratio, flags, ctr_clamp,
               G.stream_of(rois))
        return (None, gd) + (None,) * 7


def delta2bbox(rois, deltas, means=None, stds=None, max_shape=None, wh_ratio_clip=16 / 1000, clip_border=True,
               add_ctr_clamp=False, ctr_clamp=32, box_dim=None):
    """Apply `deltas` to `rois` (delta_xywh_sph_bbox_coder.py:164-263 / rsph :167-268).  rois (N, d), deltas
    (N, num_classes*d) -> (N, num_classes*d); `max_shape` is accepted and unused, as in the reference (:253)."""
    dim = rois.size(-1) if box_dim is None else box_dim
    assert dim in (4, 5)
    means = (0.,) * dim if means is None else means
    stds = (1.,) * dim if stds is None else stds
    num_bboxes = deltas.size(0)
    if num_bboxes == 0:
        return deltas
    num_classes = deltas.size(1) // dim
    assert rois.size(0) == num_bboxes and deltas.size(1) == num_classes * dim
    G.require_hip(rois, deltas)
    r = G.as_f32(rois.detach())
    d = G.as_f32(deltas)
    flags = (1 if clip_border else 0) | (2 if add_ctr_clamp else 0)
    max_ratio = abs(math.log(wh_ratio_clip))
    return _DecodeFunction.apply(r, d, tuple(means), tuple(stds), num_classes, dim, float(max_ratio), flags,
                                 float(ctr_clamp))


class _DeltaSphCoderBase:
    box_dim = 4

    def __init__(self, target_means=None, target_stds=None, clip_border=True, add_ctr_clamp=False, ctr_clamp=32):
        self.means = (0.,) * self.box_dim if target_means is None else target_means
        self.stds = (1.,) * self.box_dim if target_stds is None else target_stds
        self.clip_border = clip_border
        self.add_ctr_clamp = add_ctr_clamp
        self.ctr_clamp = ctr_clamp

    def encode(self, bboxes, gt_bboxes):
        assert bboxes.size(0) == gt_bboxes.size(0)
        assert bboxes.size(-1) == gt_bboxes.size(-1) == self.box_dim
        return bbox2delta(bboxes, gt_bboxes, self.means, self.stds)

    def decode(self, bboxes, pred_bboxes, max_shape=None, wh_ratio_clip=16 / 1000):
        assert pred_bboxes.size(0) == bboxes.size(0)
        if pred_bboxes.ndim == 3:
            assert pred_bboxes.size(1) == bboxes.size(1)
        if pred_bboxes.ndim == 3:
            # batched (B, N, d) anchors with (B, N, num_classes * d) deltas.  The reference's decode raises here
            # (`raise NotImplemented(...)`, delta_xywh_sph_bbox_coder.py:104) and its delta2bbox mis-reads a 3-D input
            # (num_classes = N // 4, :226); boxes are independent, so the batch is decoded as one (B * N, ...) launch.
            b, n = pred_bboxes.shape[:2]
            flat = delta2bbox(bboxes.reshape(b * n, -1), pred_bboxes.reshape(b * n, -1), self.means, self.stds, max_shape,
                              wh_ratio_clip, self.clip_border, self.add_ctr_clamp, self.ctr_clamp, box_dim=self.box_dim)
            return flat.reshape(b, n, -1)
        if pred_bboxes.ndim != 2:
            raise ValueError(f'decode expects (N, C*d) or (B, N, C*d) deltas, got {tuple(pred_bboxes.shape)}')
        return delta2bbox(bboxes, pred_bboxes, self.means, self.stds, max_shape, wh_ratio_clip, self.clip_border,
                          self.add_ctr_clamp, self.ctr_clamp, box_dim=self.box_dim)


@BBOX_CODERS.register_module(force=True)
class DeltaXYWHSphBBoxCoder(_DeltaSphCoderBase):
    """(theta, phi, alpha, beta) <-> (d_theta, d_phi, d_alpha, d_beta); delta_xywh_sph_bbox_coder.py:10-113."""
    box_dim = 4


@BBOX_CODERS.register_module(force=True)
class DeltaXYWHASphBBoxCoder(_DeltaSphCoderBase):
    """(theta, phi, alpha, beta, gamma) <-> five deltas, the fifth in radians; delta_xywha_rsph_bbox_coder.py:10-113."""
    box_dim = 5

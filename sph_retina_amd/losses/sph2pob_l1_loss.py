"""`Sph2PobL1Loss` — L1 on OBB deltas of the Sph2Pob planar boxes (reference sphdet/losses/sph2pob_l1_loss.py:9-88:
`@Sph2PobTransfrom() class Sph2PobL1Loss(L1Loss)`).

Forward = two launches: `sph2pob_transform_f32(..., jitter=1)` (spherical jitter, Sph2Pob standard transform, rotated
jitter) and `sph2pob_obb_l1_fwd_f32` (bbox2delta on the planar boxes, |.|, weight); backward = `sph2pob_obb_l1_bwd_f32`
and `sph2pob_transform_bwd_f32`.  The reduction follows mmdet's `weight_reduce_loss`
(mmdet/models/losses/utils.py:30-59) on the (n, 5) element losses.
The reference's constructor stops in `pdb.set_trace()` (:24); that line is not reproduced.
"""
import ctypes

import torch
import torch.nn as nn

from .. import _torch_glue as G
from ..iou.sph_iou_api import _transform
from ..registry import LOSSES

_F32_EPS = float(torch.finfo(torch.float32).eps)


class _ObbL1Function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weight, scale, flags):
        p, t = G.as_f32(pred), G.as_f32(target)
        w = G.as_f32(weight.detach()) if weight is not None else None
        n = p.size(0)
        out = torch.empty((n, 5), dtype=torch.float32, device=p.device)
        if n:
            G.call('sph2pob_obb_l1_fwd_f32', p.device, G.ptr(p), G.ptr(t), G.ptr(w), ctypes.c_float(scale), G.ptr(out),
                   ctypes.c_int64(n), flags, G.stream_of(p))
        ctx.save_for_backward(p, t, w)
        ctx.meta = (scale, flags)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        p, t, w = ctx.saved_tensors
        scale, flags = ctx.meta
        g = G.as_f32(grad_out)
        gp, gt = torch.empty_like(p), torch.empty_like(t)
        if p.size(0):
            G.call('sph2pob_obb_l1_bwd_f32', p.device, G.ptr(p), G.ptr(t), G.ptr(w), G.ptr(g), ctypes.c_float(scale),
                   G.ptr(gp), G.ptr(gt), ctypes.c_int64(p.size(0)), flags, G.stream_of(p))
        return gp, gt, None, None, None


def _reduce(loss, reduction, avg_factor):
    """weight_reduce_loss (mmdet/models/losses/utils.py:30-59) after the weight has been applied in the kernel."""
    if avg_factor is None:
        if reduction == 'mean':
            return loss.mean()
        return loss.sum() if reduction == 'sum' else loss
    if reduction == 'mean':
        return loss.sum() / (avg_factor + _F32_EPS)
    if reduction != 'none':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


@LOSSES.register_module(force=True)
class Sph2PobL1Loss(nn.Module):
    """Sph2PobL1Loss(encode=True, swap=False, angle_modifier='original', reduction='mean', loss_weight=1.0)
    .forward(pred, target, weight=None, avg_factor=None, reduction_override=None) on SPHERICAL boxes (n, 4|5) deg."""

    def __init__(self, encode=True, swap=False, angle_modifier='original', reduction='mean', loss_weight=1.0):
        assert angle_modifier in ['original', 'modulus']
        super().__init__()
        self.encode = encode
        self.swap = swap
        self.angle_modifier = angle_modifier
        self.reduction = reduction
        self.loss_weight = loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        box_version = target.size(-1)
        G.require_hip(pred, target)
        planar_pred, planar_target = _transform('standard', pred, target, 'rad', 'arc', 'equator', jitter=True)
        if weight is not None and weight.dim() > 1 and box_version == 4:   # sph2pob_transform.py:32-34
            weight = torch.cat([weight, weight.mean(-1, keepdim=True)], dim=-1)
        if weight is not None and weight.dim() == 1:
            weight = weight[:, None].expand(-1, 5)
        if planar_target.numel() == 0:                                      # smooth_l1_loss.py:47-48
            return planar_pred.sum() * 0
        flags = (1 if self.encode else 0) | (2 if self.swap else 0) | (4 if self.angle_modifier == 'modulus' else 0)
        loss = _ObbL1Function.apply(planar_pred, planar_target, weight, float(self.loss_weight), flags)
        return _reduce(loss, reduction, avg_factor)

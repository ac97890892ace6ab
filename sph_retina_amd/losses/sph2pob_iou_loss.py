"""`Sph2PobIoULoss` — the reference's registered IoU-family regression loss for spherical boxes
(sphdet/losses/sph2pob_iou_loss.py:218-220 = OBBIoULoss (:16-58) wrapped by Sph2PobTransfrom), fused end to end:

    forward : ONE kernel  (jitter -> sph2pob_standard -> jitter -> rotated IoU -> IoU|GIoU|DIoU|CIoU -> x weight)
              + a deterministic two-pass sum for 'mean' / 'sum'
    backward: ONE kernel  (recomputes the forward in registers, closed-form adjoint down to the spherical inputs)

Constructor / call signature, assertion behaviour, weight handling ((n,) or (n, box_dim) weights averaged per box,
sph2pob_iou_loss.py:43-48), `avg_factor` semantics (mmdet/models/losses/utils.py:47-58) and `reduction_override`
follow the reference.  Difference: the reference's all-zero-weight shortcut (:36-39) needs a device->host sync
(`torch.any`); here zero weights simply produce a zero loss with zero gradients on the normal path.
"""
import torch
import torch.nn as nn

from .. import _torch_glue as G
from ..registry import LOSSES

LOSS_MODES = {'iou': 0, 'giou': 1, 'diou': 2, 'ciou': 3}


def _mode_code(mode):
    """loss mode | arithmetic flag (reference-order transform when set_arithmetic('reference') is active)."""
    return LOSS_MODES[mode] | (G.FLAG_REFERENCE_ORDER if G.get_arithmetic() == 'reference' else 0)
_F32_EPS = float(torch.finfo(torch.float32).eps)


def _f32c(t):
    """The tensor itself when it already is contiguous fp32 (the common case: no detach / copy objects on the hot
    path), else a contiguous fp32 copy."""
    return t if t.dtype is torch.float32 and t.is_contiguous() else G.as_f32(t.detach())


class _Sph2PobLossFunction(torch.autograd.Function):
    """(pred, target[, weight]) -> scale * weighted element losses (reduce=False) or scale * their sum (reduce=True).
    One node in the autograd graph: `scale` carries loss_weight AND the 1 / n | 1 / (avg_factor + eps) of
    weight_reduce_loss, so 'mean' costs no extra torch op (and no extra backward node).
    When a gradient will be asked for (pred or target requires grad) the forward launch is the fused one
    (`sph2pob_loss_fwd_grad_f32`): the backward kernel has to recompute the whole forward anyway, so loss and gradients for
    an upstream gradient of 1 come out of ONE pass over the boxes and torch's backward only scales the stashed gradients
    (`sph2pob_loss_grad_scale_f32`).  Without a gradient in sight (evaluation) the plain forward kernels run."""

    @staticmethod
    def forward(ctx, pred, target, weight, mode_c, eps, scale, reduce):
        G.require_hip(pred, target)
        n, dim = pred.shape
        p, t = _f32c(pred), _f32c(target)
        w = _f32c(weight) if weight is not None else None
        wd = 0 if w is None else (1 if w.dim() == 1 else w.size(1))
        wp = w.data_ptr() if w is not None and n else None
        dev = p.device
        stream = G.raw_stream_of(dev)
        need_p, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        out = torch.empty(() if reduce else (n,), dtype=torch.float32, device=dev)
        if need_p or need_t:
            gp = torch.empty_like(p)
            gt = torch.empty_like(t) if need_t else None
            ws = G.loss_sum_workspace(dev, n) if reduce else None
            G.call('sph2pob_loss_fwd_grad_f32', dev, p.data_ptr() if n else None, t.data_ptr() if n else None, wp, wd, scale,
                   None if reduce else (out.data_ptr() if n else None), out.data_ptr() if reduce else None,
                   ws.data_ptr() if reduce else None, gp.data_ptr() if n else None, gt.data_ptr() if need_t and n else None,
                   n, dim, mode_c, eps, stream)
            ctx.save_for_backward(gp, gt, p, t, w)
            ctx.first = True
        elif reduce:
            G.call('sph2pob_loss_fwd_sum_f32', dev, p.data_ptr() if n else None, t.data_ptr() if n else None, wp, wd, scale,
                   out.data_ptr(), G.loss_sum_workspace(dev, n).data_ptr(), n, dim, mode_c, eps, stream)
        elif n:
            G.call('sph2pob_loss_fwd_f32', dev, p.data_ptr(), t.data_ptr(), wp, wd, scale, out.data_ptr(), None, n, dim,
                   mode_c, eps, stream)
        ctx.meta = (reduce, need_p, need_t, pred.dtype, target.dtype, mode_c, eps, scale, wd)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        gp, gt, p, t, w = ctx.saved_tensors
        reduce, need_p, need_t, pdt, tdt, mode_c, eps, scale, wd = ctx.meta
        n, dim = gp.shape
        g = _f32c(grad_out)
        stream = G.raw_stream_of(gp.device)
        if not ctx.first:
            # a second backward through the same node (retain_graph=True): the stashes were scaled in place and handed to
            # autograd by the first one — recompute with the two-pass backward kernel
            ngp, ngt = torch.empty_like(p), (torch.empty_like(t) if need_t else None)
            if n:
                G.call('sph2pob_loss_bwd_f32', p.device, p.data_ptr(), t.data_ptr(), w.data_ptr() if w is not None else None, wd,
                       g.data_ptr(), 0 if reduce else 1, scale, ngp.data_ptr(), ngt.data_ptr() if need_t else None, n, dim,
                       mode_c, eps, stream)
            return ((ngp if pdt is torch.float32 else ngp.to(pdt)) if need_p else None,
                    (ngt if tdt is torch.float32 else ngt.to(tdt)) if need_t else None, None, None, None, None, None)
        ctx.first = False
        outs = []
        for stash, need, dt in ((gp, need_p, pdt), (gt, need_t, tdt)):
            if not need or stash is None:
                outs.append(None)
                continue
            if n:   # in place: with an upstream gradient of exactly 1 the launch returns at once, the stash IS the gradient
                G.call('sph2pob_loss_grad_scale_f32', stash.device, stash.data_ptr(), g.data_ptr(), 0 if reduce else 1,
                       stash.data_ptr(), n, dim, stream)
            outs.append(stash if dt is torch.float32 else stash.to(dt))
        return outs[0], outs[1], None, None, None, None, None


def sph2pob_iou_loss(pred, target, weight=None, mode='iou', eps=1e-6, reduction='mean', avg_factor=None,
                     loss_weight=1.0):
    """Functional form: loss_weight * weight_reduce_loss(obb_iou_loss(sph2pob(pred, target)), weight, ...)."""
    assert mode in LOSS_MODES
    if pred.dim() != 2 or pred.shape != target.shape or pred.size(1) not in (4, 5):
        raise ValueError(f'pred/target must both be (n, 4) or (n, 5), got {tuple(pred.shape)}, {tuple(target.shape)}')
    if avg_factor is not None and reduction == 'sum':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    n = pred.size(0)
    if reduction == 'none':
        return _Sph2PobLossFunction.apply(pred, target, weight, _mode_code(mode), float(eps), float(loss_weight), False)
    assert reduction in ('mean', 'sum')
    if reduction == 'sum':
        scale = float(loss_weight)
    elif avg_factor is None:
        if n == 0:   # torch: mean of an empty tensor is nan
            return _Sph2PobLossFunction.apply(pred, target, weight, _mode_code(mode), float(eps), 1.0, True) * float('nan')
        scale = float(loss_weight) / n
    elif isinstance(avg_factor, torch.Tensor):   # a device scalar (e.g. an all-reduced positive count): no host sync
        total = _Sph2PobLossFunction.apply(pred, target, weight, _mode_code(mode), float(eps), float(loss_weight), True)
        return total / (avg_factor + _F32_EPS)
    else:
        scale = float(loss_weight) / (float(avg_factor) + _F32_EPS)
    return _Sph2PobLossFunction.apply(pred, target, weight, _mode_code(mode), float(eps), scale, True)


class OBBIoULoss(nn.Module):
    """Same constructor / forward signature as the reference's OBBIoULoss (sph2pob_iou_loss.py:16-58); `forward`
    takes SPHERICAL boxes (the Sph2Pob transform is inside the fused kernels)."""

    def __init__(self, mode='iou', eps=1e-6, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert mode in ['iou', 'giou', 'diou', 'ciou']
        self.mode = mode
        self.eps = eps
        self.reduction = reduction
        self.loss_weight = loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if weight is not None and weight.dim() > 1:
            assert weight.shape == pred.shape
        return sph2pob_iou_loss(pred, target, weight, mode=self.mode, eps=self.eps, reduction=reduction,
                                avg_factor=avg_factor, loss_weight=self.loss_weight)


@LOSSES.register_module(force=True)
class Sph2PobIoULoss(OBBIoULoss):
    """dict(type='Sph2PobIoULoss', mode='ciou', loss_weight=1.0) — reference sph2pob_iou_loss.py:218-235."""
    pass


@LOSSES.register_module(force=True)
class SphIoULoss(OBBIoULoss):
    """The reference's SphIoULoss (sph2pob_iou_loss.py:238-292) cannot be constructed at HEAD (its default
    iou_calculator is rejected by its own assert) and only implements mode='iou' through the Sph2Pob IoU.  Served
    here by the same fused kernels; `iou_calculator` accepts both spellings of the Sph2Pob backend."""

    def __init__(self, mode='iou', iou_calculator='sph2pob_standard_iou', eps=1e-6, reduction='mean', loss_weight=1.0):
        assert iou_calculator in ['sph2pob_standard', 'sph2pob_standard_iou'], \
            "only the Sph2Pob calculator is on the MI355X hot path ('sph'/'fov' closed forms: SURVEY §8f-4)"
        super().__init__(mode=mode, eps=eps, reduction=reduction, loss_weight=loss_weight)


@LOSSES.register_module(force=True)
class SphIoULossLegacy(nn.Module):
    """`@Sph2PobTransfrom() class SphIoULossLegacy(RotatedIoULoss)` (reference sph2pob_iou_loss.py:199-216): mmrotate
    0.3.2's RotatedIoULoss — `-log(IoU)` ('log', default), `1 - IoU` ('linear') or `1 - IoU^2` ('square') of the
    differentiable rotated IoU clamped at `eps` — on the Sph2Pob planar boxes.  The IoU and its gradient come from the
    fused loss kernels (the `iou` mode, pinned by the reference-generated loss fixtures); the scalar post-map is three
    element-wise torch ops.  mmrotate is absent here: its post-map is restated from the published source — parity of
    that part is UNPINNED."""

    def __init__(self, linear=False, eps=1e-6, reduction='mean', loss_weight=1.0, mode='log'):
        super().__init__()
        assert mode in ['linear', 'square', 'log']
        self.mode = 'linear' if linear else mode
        self.eps = eps
        self.reduction = reduction
        self.loss_weight = loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if weight is not None and weight.dim() > 1:
            if target.size(-1) == 4:   # Sph2PobTransfrom widens (n, 4) weights with their mean (sph2pob_transform.py:32-34)
                weight = torch.cat([weight, weight.mean(-1, keepdim=True)], dim=-1)
            weight = weight.mean(-1)
        if weight is not None and not torch.any(weight > 0) and reduction != 'none':
            return (pred * weight[:, None]).sum()
        ious = 1.0 - sph2pob_iou_loss(pred, target, mode='iou', reduction='none')
        ious = ious.clamp(min=self.eps)
        loss = 1 - ious if self.mode == 'linear' else (1 - ious ** 2 if self.mode == 'square' else -ious.log())
        if weight is not None:
            loss = loss * weight
        if avg_factor is None:
            loss = loss.mean() if reduction == 'mean' else (loss.sum() if reduction == 'sum' else loss)
        elif reduction == 'mean':
            loss = loss.sum() / (avg_factor + _F32_EPS)
        elif reduction != 'none':
            raise ValueError('avg_factor can not be used with reduction="sum"')
        return self.loss_weight * loss

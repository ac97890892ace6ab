"""Sph2Pob IoU operators — same names, positional order and defaults as the reference's
sphdet/iou/sph_iou_api.py:91-98, served by fused HIP kernels instead of ~100 torch launches + mmcv.

Behaviour kept from the reference (`_sph2pob_iou_auxiliary`, sph_iou_api.py:48-86):
  * AssertionError for bad mode / calculator / rbb_edge / rbb_angle;
  * rows = bboxes1, cols = bboxes2; aligned output shape (rows,), pairwise (rows, cols);
  * empty input -> shape (rows, 1) if is_aligned else (rows, cols) (zeros here; uninitialised there);
  * inputs are never mutated (reference tests/test_all_ious.py:322-332); output is a new fp32 tensor on the
    inputs' device, enqueued on the current stream (no host synchronisation anywhere).
Differences: `calculator='diff'` (a NameError in the reference at HEAD, sph_iou_api.py:14,81) returns the same
values as 'common'; the m*n expanded pair tensors are never materialised.
"""
import ctypes

import torch

from .. import _torch_glue as G


def _options(mode, calculator, rbb_edge, rbb_angle, legacy=False):
    assert mode in ['iou', 'iof']
    assert calculator in ['common', 'diff']
    assert rbb_edge in ['arc', 'chord', 'tangent']
    if not legacy:
        assert rbb_angle in ['equator', 'project']
    return G.MODES[mode], G.EDGES[rbb_edge], G.ANGLES.get(rbb_angle, 0)


def _sph2pob_iou_auxiliary(bboxes1, bboxes2, variant, mode, is_aligned, calculator, rbb_edge, rbb_angle):
    mode_c, edge_c, angle_c = _options(mode, calculator, rbb_edge, rbb_angle, legacy=(variant == 'legacy'))
    rows = bboxes1.size(0)
    cols = bboxes2.size(0)
    if rows * cols == 0:
        return bboxes1.new_zeros((rows, 1)) if is_aligned else bboxes1.new_zeros((rows, cols))
    G.require_hip(bboxes1, bboxes2)
    dim = bboxes1.size(1)
    if bboxes2.size(1) != dim or dim not in (4, 5):
        raise ValueError(f'boxes must both be (n, 4) BFoV or (n, 5) RBFoV, got {tuple(bboxes1.shape)} and '
                         f'{tuple(bboxes2.shape)}')
    if variant in ('legacy', 'sph_iou', 'fov_iou') and dim == 5:
        # the reference raises ValueError from torch.chunk(…, 4) on 5 columns: sph2pob_legacy.py:52-53
        raise ValueError(f'{variant} supports BFoV (n, 4) boxes only')
    if (calculator == 'diff' and is_aligned and variant == 'standard' and mode == 'iou' and rbb_edge == 'arc'
            and rbb_angle == 'equator' and torch.is_grad_enabled() and (bboxes1.requires_grad or bboxes2.requires_grad)):
        # the differentiable route the reference intended for SphIoULoss (sph2pob_iou_loss.py:292 passes
        # calculator='diff'): IoU = 1 - (IoU-mode loss element), gradients from the fused loss backward kernel
        from ..losses.sph2pob_iou_loss import sph2pob_iou_loss
        return 1.0 - sph2pob_iou_loss(bboxes1, bboxes2, mode='iou', reduction='none')
    b1, b2 = G.as_f32_nograd(bboxes1), G.as_f32_nograd(bboxes2)
    if is_aligned:
        assert rows == cols
        out = torch.empty((rows,), dtype=torch.float32, device=b1.device)
        G.call('sph2pob_iou_aligned_f32', b1.device, G.ptr(b1), G.ptr(b2), G.ptr(out), rows, dim,
               G.VARIANTS[variant], mode_c, edge_c, angle_c, G.raw_stream_of(b1.device))
    else:
        out = torch.empty((rows, cols), dtype=torch.float32, device=b1.device)
        G.call('sph2pob_iou_pairwise_f32', b1.device, G.ptr(b1), rows, G.ptr(b2),
               cols, G.ptr(out), dim, G.VARIANTS[variant], mode_c, edge_c, angle_c,
               G.raw_stream_of(b1.device))
    # fp64 / fp16 / bf16 boxes get their dtype back (computed in fp32); integer boxes get fp32 IoUs
    return out if bboxes1.dtype == torch.float32 or not bboxes1.is_floating_point() else out.to(bboxes1.dtype)


def sph_iou(bboxes1, bboxes2, mode='iou', is_aligned=False, calculator='diff'):
    """Sph-IoU (AAAI-2020 closed form), reference sph_iou_api.py:128-150: jitter -> sph_iou_aligned -> clamp."""
    assert mode in ['iou']
    return _sph2pob_iou_auxiliary(bboxes1, bboxes2, 'sph_iou', 'iou', is_aligned, 'common', 'arc', 'equator')


def fov_iou(bboxes1, bboxes2, mode='iou', is_aligned=False, calculator='diff'):
    """FoV-IoU (arXiv 2202.03176 closed form), reference sph_iou_api.py:155-175."""
    assert mode in ['iou']
    return _sph2pob_iou_auxiliary(bboxes1, bboxes2, 'fov_iou', 'iou', is_aligned, 'common', 'arc', 'equator')


def unbiased_iou(bboxes1, bboxes2, mode='iou', is_aligned=False):
    """Unbiased IoU — exact area of the intersection of two spherical rectangles (BFoV or RBFoV), reference
    sph_iou_api.py:103-126 over unbiased_iou_bfov.py / unbiased_iou_rbfov.py (numpy on the CPU there; one fp64 HIP
    kernel here).  `set_arithmetic('reference')` additionally reproduces numpy's fp32 roundings on fp32 inputs."""
    assert mode in ['iou']
    return _sph2pob_iou_auxiliary(bboxes1, bboxes2, 'unbiased', 'iou', is_aligned, 'common', 'arc', 'equator')


def naive_iou(bboxes1, bboxes2, mode='iou', is_aligned=False, box_formator='sph2pix'):
    """Naive IoU — planar IoU of the boxes drawn in ERP pixels (reference sph_iou_api.py:179-197): axis-aligned for
    BFoV (mmcv bbox_overlaps), rotated for RBFoV (mmcv box_iou_rotated); no jitter, no clamp.  `box_formator`: 'sph2pix'
    (proportional widths) or 'sph2tan' (w = 2R tan(alpha / 2): box_formator.py:98-106)."""
    assert mode in ['iou']
    assert box_formator in ['sph2pix', 'sph2tan']   # Sph2PlanarBoxTransform.__init__ (box_formator.py:163)
    return _sph2pob_iou_auxiliary(bboxes1, bboxes2, 'naive' if box_formator == 'sph2pix' else 'naive_tan', 'iou', is_aligned, 'common',
                                  'arc', 'equator')


def sph2pob_legacy_iou(bboxes1, bboxes2, mode='iou', is_aligned=False, calculator='common', rbb_edge='arc'):
    return _sph2pob_iou_auxiliary(bboxes1, bboxes2, 'legacy', mode, is_aligned, calculator, rbb_edge, None)


def sph2pob_standard_iou(bboxes1, bboxes2, mode='iou', is_aligned=False, calculator='common', rbb_edge='arc',
                         rbb_angle='equator'):
    return _sph2pob_iou_auxiliary(bboxes1, bboxes2, 'standard', mode, is_aligned, calculator, rbb_edge, rbb_angle)


def sph2pob_efficient_iou(bboxes1, bboxes2, mode='iou', is_aligned=False, calculator='common', rbb_edge='arc',
                          rbb_angle='equator'):
    return _sph2pob_iou_auxiliary(bboxes1, bboxes2, 'efficient', mode, is_aligned, calculator, rbb_edge,
                                  rbb_angle)


# ---------------------------------------------------------------------------------------------------------
# Transforms: sph2pob_{standard,efficient,legacy}(sph_gt, sph_pred, rbb_angle_version='deg', ...)
# (sphdet/iou/sph2pob_standard.py:8, sph2pob_efficient.py:9, sph2pob_legacy.py:8) -> two (n, 5) planar boxes.
class _Sph2PobTransformFunction(torch.autograd.Function):
    """(sph_gt, sph_pred) -> planar boxes (n, 5) x 2 in radians; backward = `sph2pob_transform_bwd_f32`."""

    @staticmethod
    def forward(ctx, sph_gt, sph_pred, variant, rbb_edge, rbb_angle, jitter):
        n, dim = sph_gt.shape
        b1, b2 = G.as_f32(sph_gt.detach()), G.as_f32(sph_pred.detach())
        o1 = torch.empty((n, 5), dtype=torch.float32, device=b1.device)
        o2 = torch.empty((n, 5), dtype=torch.float32, device=b1.device)
        if n:
            G.call('sph2pob_transform_f32', b1.device, G.ptr(b1), G.ptr(b2), G.ptr(o1), G.ptr(o2), ctypes.c_int64(n),
                   dim, G.VARIANTS[variant], G.EDGES[rbb_edge], G.ANGLES.get(rbb_angle, 0), int(bool(jitter)),
                   G.stream_of(b1))
        ctx.save_for_backward(b1, b2)
        ctx.meta = (variant, rbb_edge, rbb_angle, jitter, sph_gt.dtype, sph_pred.dtype)
        return o1, o2

    @staticmethod
    def backward(ctx, g1, g2):
        b1, b2 = ctx.saved_tensors
        variant, rbb_edge, rbb_angle, jitter, dt1, dt2 = ctx.meta
        n, dim = b1.shape
        gb1, gb2 = torch.empty_like(b1), torch.empty_like(b2)
        if n and (variant == 'legacy' or rbb_angle == 'project'):
            # no closed-form adjoint for these two: forward-mode differentiation of the reference-order transform
            g1, g2 = G.as_f32(g1), G.as_f32(g2)
            G.call('sph2pob_transform_bwd_general_f32', b1.device, G.ptr(b1), G.ptr(b2), G.ptr(g1), G.ptr(g2), G.ptr(gb1),
                   G.ptr(gb2), ctypes.c_int64(n), dim, G.VARIANTS[variant] & 0xff, G.EDGES[rbb_edge],
                   G.ANGLES.get(rbb_angle, 0), int(bool(jitter)), G.stream_of(b1))
        elif n:
            g1, g2 = G.as_f32(g1), G.as_f32(g2)
            G.call('sph2pob_transform_bwd_f32', b1.device, G.ptr(b1), G.ptr(b2), G.ptr(g1), G.ptr(g2), G.ptr(gb1),
                   G.ptr(gb2), ctypes.c_int64(n), dim, G.VARIANTS[variant], G.EDGES[rbb_edge], int(bool(jitter)),
                   G.stream_of(b1))
        return gb1.to(dt1), gb2.to(dt2), None, None, None, None


def _transform(variant, sph_gt, sph_pred, rbb_angle_version, rbb_edge, rbb_angle, jitter=False):
    assert rbb_angle_version in ['deg', 'rad']
    assert rbb_edge in ['arc', 'chord', 'tangent']
    if variant != 'legacy':
        assert rbb_angle in ['equator', 'project']
    G.require_hip(sph_gt, sph_pred)
    n, dim = sph_gt.shape
    if variant == 'legacy' and dim == 5:
        raise ValueError('sph2pob_legacy supports BFoV (n, 4) boxes only')
    o1, o2 = _Sph2PobTransformFunction.apply(sph_gt, sph_pred, variant, rbb_edge, rbb_angle, jitter)
    if rbb_angle_version == 'deg':
        o1 = torch.cat([o1[:, :4], torch.rad2deg(o1[:, 4:])], dim=1)
        o2 = torch.cat([o2[:, :4], torch.rad2deg(o2[:, 4:])], dim=1)
    return o1, o2


def sph2pob_standard(sph_gt, sph_pred, rbb_angle_version='deg', rbb_edge='arc', rbb_angle='equator'):
    return _transform('standard', sph_gt, sph_pred, rbb_angle_version, rbb_edge, rbb_angle)


def sph2pob_efficient(sph_gt, sph_pred, rbb_angle_version='deg', rbb_edge='arc', rbb_angle='equator'):
    return _transform('efficient', sph_gt, sph_pred, rbb_angle_version, rbb_edge, rbb_angle)


def sph2pob_legacy(sph_gt, sph_pred, rbb_angle_version='deg', rbb_edge='arc', rbb_angle=None):
    return _transform('legacy', sph_gt, sph_pred, rbb_angle_version, rbb_edge, rbb_angle)


# ---------------------------------------------------------------------------------------------------------
# The reference exposes its two jitter helpers as importable functions operating IN PLACE on clones made by
# the caller (sph_iou_api.py:222-260; used by sphdet/losses/sph2pob_transform.py:28,30).  The fused kernels
# apply both jitters in registers; these torch versions exist for callers that import the helpers directly.
def jiter_rotated_bboxes(bboxes1, bboxes2):
    eps = 1e-4 * 1.2345678
    e1 = bboxes1.new_tensor([eps, eps, 2 * eps, 2 * eps, eps])
    e2 = bboxes1.new_tensor([2 * eps, 2 * eps, eps, eps, 5 * eps])
    cols = [0, 2, 3, 4]
    similar = ((bboxes1[:, cols] - bboxes2[:, cols]).abs() < eps).any(dim=1, keepdim=True)
    bboxes1.add_(similar * e1)
    bboxes2.add_(similar * e2)
    eps = 1e-3 * 1.2345678
    close = (bboxes1[:, 4] - bboxes2[:, 4]).abs() < eps
    bboxes1[:, 4].add_(close * eps)
    bboxes2[:, 4].add_(close * (2 * eps))
    pi = torch.pi
    bboxes1[:, 2:4].clamp_(min=2 * eps / 10)
    bboxes2[:, 2:4].clamp_(min=eps / 10)
    bboxes1[:, 4].clamp_(min=-2 * pi + 2 * eps, max=2 * pi - eps)
    bboxes2[:, 4].clamp_(min=-2 * pi + eps, max=2 * pi - 2 * eps)
    return bboxes1, bboxes2


def jiter_spherical_bboxes(bboxes1, bboxes2):
    eps = 1e-4 * 1.2345678
    similar = ((bboxes1 - bboxes2).abs() < eps).any(dim=1, keepdim=True)
    bboxes1.sub_(similar * (2 * eps))
    bboxes2.add_(similar * eps)
    pi = 180
    bboxes1[:, 0].clamp_(2 * eps, 2 * pi - eps)
    bboxes1[:, 1:4].clamp_(2 * eps, pi - eps)
    bboxes2[:, 0].clamp_(eps, 2 * pi - 2 * eps)
    bboxes2[:, 1:4].clamp_(eps, pi - 2 * eps)
    if bboxes1.size(1) == 5:
        bboxes2[:, 4].clamp_(-2 * pi + 2 * eps, 2 * pi - 2 * eps)
    return bboxes1, bboxes2

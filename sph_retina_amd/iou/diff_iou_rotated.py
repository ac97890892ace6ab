"""Planar rotated-rectangle IoU on GIVEN planar boxes — the two spellings the reference uses for it:

    box_iou_rotated(bboxes1, bboxes2, mode='iou', aligned=False, clockwise=True)   mmcv.ops (call sites
        sphdet/iou/sph_iou_api.py:79, :193)
    diff_iou_rotated_2d(box1, box2)   (B, N, 5) x (B, N, 5) -> (B, N); sphdet/iou/diff_iou_rotated.py:325-343, the vendored
        value-equivalent the reference's own script calls directly (tests/test_all_ious.py:22-24)

Both are served by `sph2pob_planar_iou_f32` (boundary-integral clip; boxes (x, y, w, h, a) with the angle in radians).
Values only: the differentiable use of this op inside the reference — the IoU-family losses — is fused end to end in
`sph_retina_amd.losses` (forward and hand-derived backward in one kernel each), so a tensor that requires grad is
rejected here instead of silently returning a constant.
"""
import torch

from .. import _torch_glue as G


def _planar(b1, b2, aligned, mode):
    assert mode in ['iou', 'iof']
    G.require_hip(b1, b2)
    if torch.is_grad_enabled() and (b1.requires_grad or b2.requires_grad):
        raise RuntimeError('the stand-alone planar IoU is forward-only; for gradients use sph_retina_amd.losses '
                           '(Sph2PobIoULoss / obb losses: fused forward + backward kernels)')
    if b1.size(-1) != 5 or b2.size(-1) != 5:
        raise ValueError(f'planar boxes are (n, 5) = (x, y, w, h, angle), got {tuple(b1.shape)}, {tuple(b2.shape)}')
    m, n = b1.size(0), b2.size(0)
    if aligned:
        assert m == n
    out = torch.empty((m,) if aligned else (m, n), dtype=torch.float32, device=b1.device)
    if m and n:
        p1, p2 = G.as_f32(b1.detach()), G.as_f32(b2.detach())
        G.call('sph2pob_planar_iou_f32', p1.device, p1.data_ptr(), m, p2.data_ptr(), n, out.data_ptr(), int(bool(aligned)),
               G.MODES[mode], G.raw_stream_of(p1.device))
    return out


def box_iou_rotated(bboxes1, bboxes2, mode='iou', aligned=False, clockwise=True):
    """mmcv.ops.box_iou_rotated: rows = bboxes1.  `clockwise` only flips the sign convention of both angles, which leaves
    the IoU of a pair unchanged; it is accepted for signature compatibility."""
    return _planar(bboxes1, bboxes2, aligned, mode)


def diff_iou_rotated_2d(box1, box2):
    """(B, N, 5), (B, N, 5) -> (B, N) aligned IoUs."""
    assert box1.dim() == 3 and box1.shape == box2.shape and box1.size(-1) == 5
    b, n = box1.shape[:2]
    return _planar(box1.reshape(b * n, 5), box2.reshape(b * n, 5), True, 'iou').reshape(b, n)

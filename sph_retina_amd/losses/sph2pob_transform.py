"""`Sph2PobTransfrom` — class decorator of the reference (sphdet/losses/sph2pob_transform.py:11-37) that rewrites
an OBB loss's `forward` so that spherical boxes are transformed to planar oriented boxes first.

Here the three steps the reference runs as ~70 torch launches (spherical jitter, sph2pob transform, rotated jitter,
:26-30) are ONE kernel launch (`sph2pob_transform_f32(..., jitter=1)`).  It serves the wrapped OBB losses that are
not fused end-to-end (L1 / GD / KF bodies are mmrotate math: SURVEY §8f-3).  The returned planar boxes carry an
autograd node whose backward is ONE kernel (`sph2pob_transform_bwd_f32`, the closed-form adjoint of the transform
and of both jitters' clamp gates), so any torch OBB loss body wrapped by this decorator back-propagates to the
spherical inputs exactly like the reference's autograd does through ~70 recorded ops.
`Sph2PobIoULoss` does not use it: it is fused end to end with its own backward (sph2pob_iou_loss.py here).
"""
import functools

import torch

from ..iou.sph_iou_api import _transform


class Sph2PobTransfrom:
    def __init__(self, transform='sph2pob_standard'):
        assert transform in ['sph2pob_standard', 'sph2pob_legacy']
        self.variant = transform[len('sph2pob_'):]

    def __call__(self, cls):
        old_forward = cls.forward
        variant = self.variant

        @functools.wraps(old_forward)
        def new_forward(_self_, pred, target, weight=None, *args, **kwargs):
            box_version = target.size(-1)
            pred, target = _transform(variant, pred, target, 'rad', 'arc', 'equator', jitter=True)
            if weight is not None and weight.dim() > 1:
                if box_version == 4:
                    weight = torch.cat([weight, weight.mean(-1, keepdim=True)], dim=-1)
            return old_forward(_self_, pred, target, weight, *args, **kwargs)

        cls.forward = new_forward
        return cls

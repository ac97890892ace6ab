"""Registry-facing IoU calculator — mirrors sphdet/iou/sph_iou_calculator.py:8-113.

`SphOverlaps2D` is what mmdet's `MaxIoUAssigner` builds from
`dict(type='SphOverlaps2D', backend='sph2pob_standard_iou', box_version=4)`
(configs/retinanet/sph_retinanet_r50_fpn_120e_indoor360.py:37-40) and calls as `iou(gt_bboxes, bboxes)`.
"""
import torch

from ..registry import IOU_CALCULATORS
from .sph_iou_api import (fov_iou, naive_iou, sph2pob_efficient_iou, sph2pob_legacy_iou, sph2pob_standard_iou, sph_iou,
                          unbiased_iou)

_ALL_BACKENDS = ['unbiased_iou', 'sph2pob_standard_iou', 'sph2pob_legacy_iou', 'sph2pob_efficient_iou', 'naive_iou',
                 'fov_iou', 'sph_iou', 'kent_iou']
_HIP_BACKENDS = {'sph2pob_standard_iou': sph2pob_standard_iou, 'sph2pob_legacy_iou': sph2pob_legacy_iou,
                 'sph2pob_efficient_iou': sph2pob_efficient_iou, 'sph_iou': sph_iou, 'fov_iou': fov_iou,
                 'unbiased_iou': unbiased_iou, 'naive_iou': naive_iou}


@IOU_CALCULATORS.register_module(force=True)
class SphOverlaps2D(object):
    """2D Overlaps calculator for spherical boxes (reference: sph_iou_calculator.py:8-56)."""

    def __init__(self, backend='unbiased_iou', box_version=4):
        self.backend = backend
        self.box_version = box_version

    def __call__(self, bboxes1, bboxes2, mode='iou', is_aligned=False):
        assert bboxes1.size(-1) in [0, 4, 5, 6]
        assert bboxes2.size(-1) in [0, 4, 5, 6]
        bboxes1 = bboxes1[..., :self.box_version]  # drops a trailing score column
        bboxes2 = bboxes2[..., :self.box_version]
        with torch.no_grad():
            return sph_overlaps(bboxes1, bboxes2, mode, is_aligned, self.backend)

    def __repr__(self):
        return self.__class__.__name__ + '()'


def sph_overlaps(bboxes1, bboxes2, mode='iou', is_aligned=False, backend='unbiased_iou'):
    """Backend switch (reference: sph_iou_calculator.py:58-113); rows = bboxes1."""
    assert mode in ['iou', 'iof']
    assert backend in _ALL_BACKENDS
    rows = bboxes1.size(0)
    cols = bboxes2.size(0)
    if rows * cols == 0:
        return bboxes1.new_zeros((rows, 1)) if is_aligned else bboxes1.new_zeros((rows, cols))
    fn = _HIP_BACKENDS.get(backend)
    if fn is None:
        raise NotImplementedError(
            f"backend '{backend}' (Kent distributions) is outside the Sph2Pob hot path served by sph_retina_amd; "
            "use one of " + ', '.join(sorted(_HIP_BACKENDS)))
    return fn(bboxes1, bboxes2, mode, is_aligned)

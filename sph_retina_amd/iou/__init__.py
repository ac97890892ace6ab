from .sph_iou_api import (fov_iou, naive_iou, unbiased_iou, jiter_rotated_bboxes, jiter_spherical_bboxes, sph2pob_efficient,  # noqa: F401
                          sph_iou,
                          sph2pob_efficient_iou, sph2pob_legacy, sph2pob_legacy_iou, sph2pob_standard,
                          sph2pob_standard_iou)
from .sph_iou_calculator import SphOverlaps2D, sph_overlaps  # noqa: F401
from .diff_iou_rotated import box_iou_rotated, diff_iou_rotated_2d  # noqa: F401

__all__ = ['box_iou_rotated', 'diff_iou_rotated_2d', 'SphOverlaps2D', 'sph_overlaps', 'sph2pob_standard_iou', 'sph2pob_legacy_iou', 'sph2pob_efficient_iou',
           'sph2pob_standard', 'sph2pob_efficient', 'sph2pob_legacy', 'jiter_spherical_bboxes',
           'jiter_rotated_bboxes', 'sph_iou', 'fov_iou', 'unbiased_iou', 'naive_iou']

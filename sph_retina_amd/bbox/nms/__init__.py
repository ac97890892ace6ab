from .sph_nms import PlanarNMS, SphNMS, sph_batched_nms, sph_nms_op  # noqa: F401
from .utils import multiclass_nms  # noqa: F401

__all__ = ['PlanarNMS', 'SphNMS', 'sph_batched_nms', 'sph_nms_op', 'multiclass_nms']

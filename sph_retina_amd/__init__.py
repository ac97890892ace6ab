"""sph_retina_amd — MI355X-native Sph2Pob spherical-IoU engine (drop-in for the reference's sphdet.iou /
sphdet.losses / sphdet.bbox.nms operator surface; kernels in csrc/, C ABI in include/sph2pob_hip.h)."""
from . import _lib  # noqa: F401
from ._torch_glue import get_arithmetic, set_arithmetic  # noqa: F401
from .iou import (SphOverlaps2D, sph2pob_efficient_iou, sph2pob_legacy_iou, sph2pob_standard_iou,  # noqa: F401
                  sph_overlaps)

from .losses import Sph2PobIoULoss, SphIoULoss  # noqa: F401,E402
from .bbox.nms import SphNMS, multiclass_nms  # noqa: F401,E402
from .bbox.assigners import SphMaxIoUAssigner  # noqa: F401,E402
from .bbox.coder import DeltaXYWHASphBBoxCoder, DeltaXYWHSphBBoxCoder  # noqa: F401,E402

__version__ = '0.1.0'

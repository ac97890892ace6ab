from .sph2pob_iou_loss import (OBBIoULoss, Sph2PobIoULoss, SphIoULoss, SphIoULossLegacy,  # noqa: F401
                               sph2pob_iou_loss)
from .sph2pob_l1_loss import Sph2PobL1Loss  # noqa: F401
from .sph2pob_transform import Sph2PobTransfrom  # noqa: F401

__all__ = ['Sph2PobIoULoss', 'SphIoULoss', 'OBBIoULoss', 'Sph2PobTransfrom', 'sph2pob_iou_loss', 'Sph2PobL1Loss', 'SphIoULossLegacy']

from . import sph2pob_mmrotate_losses as _mm  # noqa: E402  (defines Sph2PobGDLoss / Sph2PobKFLoss when mmrotate is present)
for _name in _mm.__all__:
    globals()[_name] = getattr(_mm, _name)
    __all__.append(_name)

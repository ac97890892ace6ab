"""`Sph2PobGDLoss` / `Sph2PobKFLoss` (reference sphdet/losses/sph2pob_gd_loss.py:7-27, sph2pob_kf_loss.py:8-26): mmrotate's
`GDLoss` / `KFLoss` bodies behind the `Sph2PobTransfrom` decorator.  The bodies are mmrotate code and stay mmrotate's:
when mmrotate is importable the two classes are built here exactly as the reference builds them — subclass + decorator —
so that the spherical jitter, the Sph2Pob transform and the rotated jitter in front of them, and their backward, are
the fused kernels of this package (one launch each way).  Without mmrotate (this container) the names are not defined.
"""
from ..registry import LOSSES
from .sph2pob_transform import Sph2PobTransfrom

__all__ = []

try:
    from mmrotate.models.losses import GDLoss, KFLoss
except Exception:  # mmrotate (and mmcv-full underneath it) not installed
    GDLoss = KFLoss = None

if GDLoss is not None:
    @LOSSES.register_module(force=True)
    @Sph2PobTransfrom()
    class Sph2PobGDLoss(GDLoss):
        """Gaussian-distance losses (GWD / KLD / ...) on the Sph2Pob planar boxes."""

    @LOSSES.register_module(force=True)
    @Sph2PobTransfrom()
    class Sph2PobKFLoss(KFLoss):
        """KFIoU loss on the Sph2Pob planar boxes; decoded boxes are the planar boxes themselves (:24-26)."""

        def forward(self, pred, target, *args, **kwargs):
            return super().forward(pred, target, pred_decode=target, targets_decode=pred, *args, **kwargs)

    __all__ += ['Sph2PobGDLoss', 'Sph2PobKFLoss']

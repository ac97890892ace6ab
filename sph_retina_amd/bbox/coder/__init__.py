from .delta_sph_bbox_coder import (DeltaXYWHASphBBoxCoder, DeltaXYWHSphBBoxCoder, bbox2delta,  # noqa: F401
                                   delta2bbox)

__all__ = ['DeltaXYWHSphBBoxCoder', 'DeltaXYWHASphBBoxCoder', 'bbox2delta', 'delta2bbox']

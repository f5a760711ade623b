from .bev_occ_head import BEVOCCHead2Dv2  # noqa: F401
from .mtl_occ_det_headv2 import BevFeatureSlicer, MultiTaskHeadv2  # noqa: F401

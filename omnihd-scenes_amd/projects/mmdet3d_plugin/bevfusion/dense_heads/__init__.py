from .bev_occ_head import BEVOCCHead2Dv2  # noqa: F401
from .det_anchor3d_head import Anchor3DHeadV1  # noqa: F401
from .mtl_occ_det_headv2 import BevFeatureSlicer, MultiTaskHeadv2  # noqa: F401

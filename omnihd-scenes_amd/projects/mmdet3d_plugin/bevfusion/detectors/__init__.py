from .bevf_faster_rcnn_bevdepth import BEVFUSION_depth, SE_Block  # noqa: F401
from .cam_stream_lss_bevpoolv2_depthnet import LiftSplatShoot_Depth  # noqa: F401
from .cam_stream_lss_bevpoolv2 import LiftSplatShoot  # noqa: F401
from .bevf_faster_rcnn import BEVF_FasterRCNN  # noqa: F401
from .bevf_faster_rcnn_MTL import BEVF_FasterRCNN_MTL  # noqa: F401
from .bevf_triple_temporal import BEVFusionTripleTemporal  # noqa: F401  (composition; no reference counterpart)

from .BEVCross_modal_attention import Cross_Modal_Fusion  # noqa: F401
from .rcfusion_faster_rcnn import RCFusion_FasterRCNN  # noqa: F401

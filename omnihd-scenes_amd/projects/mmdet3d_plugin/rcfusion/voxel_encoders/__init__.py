from .pillar_encoder import PillarFeatureNetV1, RadarPillarFeatureNet  # noqa: F401
from .utils import PFNLayer, PFNLayer_Radar, get_paddings_indicator  # noqa: F401

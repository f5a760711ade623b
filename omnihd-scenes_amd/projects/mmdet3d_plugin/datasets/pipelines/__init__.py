from .loading import (LoadGTDepth, LoadOccupancy_Newscenes, LoadRadarPointsMultiSweeps, RadarPoints,  # noqa: F401
                      merge_radar_sweeps)
from .transform_3d import (CustomCollect3D, NormalizeMultiviewImage, PadMultiViewImage,  # noqa: F401
                           RandomScaleImageMultiViewImage)

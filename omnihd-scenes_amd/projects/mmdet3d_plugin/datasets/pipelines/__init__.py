from .loading import (LoadGTDepth, LoadMultiViewImageFromFiles_newsc, LoadOccupancy_Newscenes,  # noqa: F401
                      LoadRadarPointsMultiSweeps, RadarPoints, merge_radar_sweeps)
from .transform_3d import (CustomCollect3D, NormalizeMultiviewImage, PadMultiViewImage,  # noqa: F401
                           RandomScaleImageMultiViewImage)

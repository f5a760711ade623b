"""MI355X-native stand-in for the hot-path slice of the reference's ``projects.mmdet3d_plugin``
package: importing it registers the same ``type=`` names (BEVFUSION_depth, FPNC, PillarFeatureNetV1,
RadarPillarFeatureNet, naiveSyncBN1d/2d/3d ...) and exposes the pooling operators under the same
module paths (``ops.bev_pool_v2.bev_pool``, ``ops.bev_pool``).  The reference's own __init__
(projects/mmdet3d_plugin/__init__.py:1-20) also pulls in datasets, BEVFormer and DD3D — outside the
hot path and not provided."""
from .bevfusion import *  # noqa: F401,F403
from .rcfusion import *  # noqa: F401,F403
from .ops.bev_pool import *  # noqa: F401,F403
from .ops import norm  # noqa: F401

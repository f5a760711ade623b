from .voxel_encoders import *  # noqa: F401,F403
from .detectors import *  # noqa: F401,F403

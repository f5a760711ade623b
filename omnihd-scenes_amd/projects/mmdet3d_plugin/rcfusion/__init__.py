from .voxel_encoders import *  # noqa: F401,F403

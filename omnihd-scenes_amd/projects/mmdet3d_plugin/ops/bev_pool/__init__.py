from .bev_pool import bev_pool  # noqa: F401

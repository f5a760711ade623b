from .fpnc import FPNC  # noqa: F401

"""omnihd_amd — host-side glue between PyTorch-ROCm tensors and libomnihd_hip.so (gfx950).

Only plumbing lives here: argument checks, stream/device hand-over and table caching.  All
arithmetic of the hot path runs in the hand-written HIP kernels behind the C ABI declared in
``include/omnihd_hip.h``.  There is NO CPU fallback: every op raises if the library or a GPU is
missing.
"""
from ._lib import lib, library_path, require_gpu  # noqa: F401
from . import ops  # noqa: F401
from .plan import BevPoolPlan, build_plan, plan_from_tables  # noqa: F401
from .pool_plan import DevicePoolPlan, build_device_plan  # noqa: F401

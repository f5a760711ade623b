"""``bev_pool_v2_ext`` — the native module the reference imports next to ``bev_pool.py``
(ops/bev_pool_v2/bev_pool.py:6; pybind definitions at ops/bev_pool_v2/src/bev_pool.cpp:106-110).

Here it is a thin veneer over libomnihd_hip.so (hand-written HIP kernels, C ABI in
include/omnihd_hip.h) with the same two function names and positional signatures.  Unlike the
reference it validates dtype/device/contiguity, launches on torch's current stream and raises on
launch errors.  CPU tensors are rejected: there is no CPU implementation.
"""
from omnihd_amd.ops import bev_pool_v2_backward, bev_pool_v2_forward  # noqa: F401

__all__ = ["bev_pool_v2_forward", "bev_pool_v2_backward"]

"""``bev_pool_ext`` — native module of the v1 pooling op (reference: ops/bev_pool/bev_pool.py:3,
pybind definitions ops/bev_pool/src/bev_pool.cpp:89-94), backed by libomnihd_hip.so."""
from omnihd_amd.ops import bev_pool_backward, bev_pool_forward  # noqa: F401

__all__ = ["bev_pool_forward", "bev_pool_backward"]

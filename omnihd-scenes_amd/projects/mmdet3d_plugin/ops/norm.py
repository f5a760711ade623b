"""``naiveSyncBN3d`` (reference: projects/mmdet3d_plugin/ops/norm.py:28-82) and its 1-D/2-D siblings
registered under the NORM_LAYERS names the configs use; implementation in omnihd_amd.mm.sync_bn."""
from omnihd_amd.mm.bricks import _register_default_norms
from omnihd_amd.mm.sync_bn import (AllReduceSum as AllReduce, NaiveSyncBatchNorm1d, NaiveSyncBatchNorm2d,  # noqa: F401
                                   NaiveSyncBatchNorm3d)

_register_default_norms()

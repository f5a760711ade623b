"""Module path of the reference's anchor head (projects/mmdet3d_plugin/bevfusion/dense_heads/det_anchor3d_head.py:
``Anchor3DHeadV1``, the detection head inside the multi-task occupancy config, bevfusion_occ.py:106).  The head is the
upstream ``Anchor3DHead`` restated in ``omnihd_amd/mm/anchor_head.py`` and registered under both names."""
from omnihd_amd.mm.anchor_head import Anchor3DHead as Anchor3DHeadV1  # noqa: F401

__all__ = ["Anchor3DHeadV1"]

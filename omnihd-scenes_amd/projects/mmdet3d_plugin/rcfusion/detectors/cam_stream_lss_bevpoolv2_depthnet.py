"""Module path of RCFusion's own copy of the Lift-Splat stream (reference:
projects/mmdet3d_plugin/rcfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py — the same classes as the bevfusion
file up to whitespace and comments, imported by rcfusion_faster_rcnn.py:10).  One implementation serves both."""
from projects.mmdet3d_plugin.bevfusion.detectors.cam_stream_lss_bevpoolv2_depthnet import (  # noqa: F401
    ASPP, CamEncode, DepthNet, LiftSplatShoot_Depth, gen_dx_bx)

__all__ = ["LiftSplatShoot_Depth", "CamEncode", "DepthNet", "ASPP", "gen_dx_bx"]

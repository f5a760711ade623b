"""Plain Lift-Splat camera stream with BEVPoolv2 — MI355X host-side mirror of
``projects/mmdet3d_plugin/bevfusion/detectors/cam_stream_lss_bevpoolv2.py`` of the reference
(``LiftSplatShoot`` :149-373, ``CamEncode`` :124-147).  This is the class BASELINE.json's first
configuration drives directly (1 camera, 256x704: ``x (B,1,256,64,176)``, ``rots (B,1,3,3)``,
``trans (B,1,3)``) and the camera stream of ``BEVF_FasterRCNN``.

It differs from ``LiftSplatShoot_Depth`` in two places only: the depth head is one 1x1 convolution
(``camencode.depthnet.{weight,bias}``) instead of the DepthNet, and the BEV encoder carries plain
``BatchNorm2d`` layers (eps 1e-5).  Geometry, rank tables and the pooling call are shared — the
reference repeats them line for line (:222-362 here vs :222-362 there) — so the whole pooling path is
the same HIP path: cached ``BevPoolPlan`` per calibration, dense tiled kernel, zero-copy ``s2c``.
State-dict keys are the reference's (``frustum``, ``camencode.depthnet.*``,
``bevencode.{0,1,3,4,6,7,9,10}.*``).  No CPU path: CPU tensors raise.

``cumsum_trick`` / ``QuickCumsum`` (:88-122) are kept for API completeness; nothing on the v2 path
calls them (``use_quickcumsum`` is a dead switch in the reference as well).
"""
import torch
from torch import nn

from projects.mmdet3d_plugin.ops.bev_pool_v2.bev_pool import bev_pool_v2  # noqa: F401  (API parity)

from .cam_stream_lss_bevpoolv2_depthnet import LiftSplatShoot_Depth, gen_dx_bx  # noqa: F401

__all__ = ["LiftSplatShoot", "CamEncode", "QuickCumsum", "cumsum_trick", "gen_dx_bx"]


def _segment_tails(ranks):
    """True at the last row of every run of equal ``ranks`` (rows already sorted by rank)."""
    tail = torch.ones(ranks.shape[0], device=ranks.device, dtype=torch.bool)
    tail[:-1] = ranks[1:] != ranks[:-1]
    return tail


def cumsum_trick(x, geom_feats, ranks):
    """Per-voxel sums of rank-sorted rows by differencing a running sum at the run tails (:88-96)."""
    tail = _segment_tails(ranks)
    running = x.cumsum(0)[tail]
    return torch.cat((running[:1], running[1:] - running[:-1])), geom_feats[tail]


class QuickCumsum(torch.autograd.Function):
    """``cumsum_trick`` with the closed-form backward: every row receives its run's gradient (:99-122)."""

    @staticmethod
    def forward(ctx, x, geom_feats, ranks):
        tail = _segment_tails(ranks)
        ctx.save_for_backward(tail)
        sums, geom = cumsum_trick(x, geom_feats, ranks)
        ctx.mark_non_differentiable(geom)
        return sums, geom

    @staticmethod
    def backward(ctx, gradx, gradgeom):
        tail, = ctx.saved_tensors
        run_of_row = torch.cumsum(tail, 0)
        run_of_row[tail] -= 1
        return gradx[run_of_row], None, None


class CamEncode(nn.Module):
    """One 1x1 convolution to D depth logits + C context channels (:124-147)."""

    def __init__(self, D, C, inputC):
        super().__init__()
        self.D, self.C = D, C
        self.depthnet = nn.Conv2d(inputC, D + C, kernel_size=1, padding=0)

    def get_depth_dist(self, x, eps=1e-20):
        return x.softmax(dim=1)

    def get_depth_feat(self, x):
        x = self.depthnet(x)
        return self.get_depth_dist(x[:, :self.D]), x[:, self.D:(self.D + self.C)]

    def forward(self, x):
        depth, feat = self.get_depth_feat(x)
        return feat, depth


class LiftSplatShoot(LiftSplatShoot_Depth):
    """Camera features (B,N,C,fH,fW) + calibration -> BEV features (B,inputC,Y,X) and the depth
    distribution (B,N,D,fH,fW).  Constructor arguments as the reference (:150-151); ``lss=True`` (the
    ResNet-18 BEV encoder of the original LSS, :39-78) is used by no NewScenes config and raises."""

    def __init__(self, lss=False, final_dim=(900, 1600), camera_depth_range=[4.0, 45.0, 1.0],
                 pc_range=[-50, -50, -5, 50, 50, 3], downsample=4, grid=3, inputC=256, camC=64):
        super().__init__(lss=lss, final_dim=final_dim, camera_depth_range=camera_depth_range, pc_range=pc_range,
                         downsample=downsample, grid=grid, inputC=inputC, camC=camC, norm_cfg=dict(type="BN"))

    def _build_camencode(self):
        return CamEncode(self.D, self.camC, self.inputC)

    def get_depth_loss(self, *a, **k):
        raise AttributeError("the plain LiftSplatShoot has no depth loss; BEVF_FasterRCNN.depth_dist_loss is the "
                             "reference's supervision for it (bevf_faster_rcnn.py:221-235)")

    get_klv_depth_loss = get_depth_loss

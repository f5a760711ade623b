"""``BEVF_FasterRCNN`` — the BEVFusion detector with the plain Lift-Splat camera stream, host-side
mirror of projects/mmdet3d_plugin/bevfusion/detectors/bevf_faster_rcnn.py (class :27-235).  Same
registry name, constructor arguments, attribute / state-dict names and loss keys.

It is ``BEVFUSION_depth`` with two substitutions, which is how the reference relates the two files
(the bodies of ``extract_feat`` / ``simple_test`` / ``forward_train`` are otherwise the same text):
the camera stream is ``LiftSplatShoot`` (1x1-convolution depth head, :54-56) and the depth
supervision takes a PRE-COMPUTED target, ``img_depth (B,N,fH,fW,1+D)`` = [min depth, D-bin
distribution], through ``depth_dist_loss`` (:221-235) instead of building the Gaussian target from a
depth map.  Everything on the hot path (voxelisation, pillar scatter, pooling plan, dual-stream
radar branch, fused BN epilogues, MFMA weight gradients) is inherited.

Not carried over: the periodic ``draw_bev_img`` debug rendering (:144-148; imports a module the
reference tree does not contain)."""
import torch
import torch.nn.functional as F

from omnihd_amd.mm import DETECTORS

from .bevf_faster_rcnn_bevdepth import BEVFUSION_depth, SE_Block  # noqa: F401
from .cam_stream_lss_bevpoolv2 import LiftSplatShoot

__all__ = ["BEVF_FasterRCNN", "SE_Block"]


@DETECTORS.register_module()
class BEVF_FasterRCNN(BEVFUSION_depth):
    def __init__(self, freeze_img=False, lss=False, lc_fusion=False, camera_stream=False,
                 camera_depth_range=[4.0, 45.0, 1.0], img_depth_loss_weight=1.0, img_depth_loss_method="kld",
                 grid=0.6, num_views=6, se=False, final_dim=(900, 1600), pc_range=[-50, -50, -5, 50, 50, 3],
                 downsample=4, imc=256, lic=384, **kwargs):
        if "norm_cfg" in kwargs:
            raise TypeError("BEVF_FasterRCNN takes no norm_cfg (reference :30-33): its fusion conv is fixed to "
                            "BN(eps=1e-3, momentum=0.01) and its camera stream to plain BatchNorm2d")
        super().__init__(freeze_img=freeze_img, lss=lss, lc_fusion=lc_fusion, camera_stream=camera_stream,
                         camera_depth_range=camera_depth_range, img_depth_loss_weight=img_depth_loss_weight,
                         img_depth_loss_method=img_depth_loss_method, grid=grid, num_views=num_views, se=se,
                         final_dim=final_dim, pc_range=pc_range, downsample=downsample, imc=imc, lic=lic,
                         norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), **kwargs)
        self.draw_interval, self.vis_time_bev = 2000, -1       # attributes the reference sets (:74-75)

    def _build_lift(self, norm_cfg, **kw):
        return LiftSplatShoot(**kw)

    def _depth_loss(self, depth_dist, img_depth):
        return self.depth_dist_loss(depth_dist, img_depth, loss_method=self.img_depth_loss_method)

    def depth_dist_loss(self, predict_depth_dist, gt_depth, loss_method="kld", img=None):
        """predict (B,N,D,H,W) vs gt (B,N,H,W,1+D) on the pixels whose min depth lies inside the camera
        depth range (:221-235).  The reference gathers the selected rows with a boolean index (one
        device->host synchronisation for the row count); the same means are taken here under a mask."""
        B, N, D, H, W = predict_depth_dist.shape
        target, min_depth = gt_depth[..., 1:].reshape(-1, D), gt_depth[..., 0].reshape(-1)
        lo, hi = self.camera_depth_range[0], self.camera_depth_range[1]
        sel = ((min_depth >= lo) & (min_depth <= hi)).unsqueeze(-1)
        pred = predict_depth_dist.float().permute(0, 1, 3, 4, 2).reshape(-1, D)
        rows = sel.sum()
        if loss_method == "kld":       # 'batchmean': summed divergence / selected rows
            kl = F.kl_div(torch.log(pred + 1e-4), target, reduction="none", log_target=False)
            return (kl * sel).sum() / rows
        if loss_method == "mse":       # mean over selected rows x D
            return (((pred - target) ** 2) * sel).sum() / (rows * D)
        raise NotImplementedError

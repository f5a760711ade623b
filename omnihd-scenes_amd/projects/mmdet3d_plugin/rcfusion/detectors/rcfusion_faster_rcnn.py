"""``RCFusion_FasterRCNN`` — the second camera + 4D-radar baseline of the reference (34.88 mAP,
README.md:208), mirror of projects/mmdet3d_plugin/rcfusion/detectors/rcfusion_faster_rcnn.py:32-225.
It shares the whole hot path with ``BEVFUSION_depth`` (same streams, same HIP operators); the
differences are the radar pillar net (``RadarPillarFeatureNet``, chosen by the config) and the fusion:
``rc_fusion='cross_attention'`` -> ``Cross_Modal_Fusion``, ``'concat'`` -> reduc_conv (+ SE)."""
import torch.nn.functional as F

from omnihd_amd.mm import DETECTORS, ConvModule
from omnihd_amd.mm.bricks import use_bev_conv
from omnihd_amd.mm.detector import MVXFasterRCNN
from projects.mmdet3d_plugin.bevfusion.detectors.bevf_faster_rcnn_bevdepth import BEVFUSION_depth, SE_Block
from projects.mmdet3d_plugin.bevfusion.detectors.cam_stream_lss_bevpoolv2_depthnet import LiftSplatShoot_Depth

from .BEVCross_modal_attention import Cross_Modal_Fusion

__all__ = ["RCFusion_FasterRCNN"]


@DETECTORS.register_module()
class RCFusion_FasterRCNN(BEVFUSION_depth):
    def __init__(self, freeze_img=False, lss=False, rc_fusion="cross_attention", camera_stream=False,
                 camera_depth_range=[4.0, 45.0, 1.0], img_depth_loss_weight=1.0, img_depth_loss_method="kld",
                 grid=0.6, num_views=6, se=False, final_dim=(900, 1600), pc_range=[-50, -50, -5, 50, 50, 3],
                 downsample=4, imc=256, lic=384, norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), **kwargs):
        MVXFasterRCNN.__init__(self, **kwargs)
        self.num_views, self.rc_fusion = num_views, rc_fusion
        self.lc_fusion = rc_fusion == "concat"
        self.img_depth_loss_weight, self.img_depth_loss_method = img_depth_loss_weight, img_depth_loss_method
        self.camera_depth_range = camera_depth_range
        self.lift, self.se = camera_stream, se
        if camera_stream:
            self.lift_splat_shot_vis = LiftSplatShoot_Depth(lss=lss, grid=grid, inputC=imc, camC=64, pc_range=pc_range,
                                                            camera_depth_range=camera_depth_range, final_dim=final_dim,
                                                            downsample=downsample, norm_cfg=norm_cfg)
        if rc_fusion == "concat":
            if se:
                self.seblock = SE_Block(lic)
            self.reduc_conv = ConvModule(lic + imc, lic, 3, padding=1, conv_cfg=None, norm_cfg=norm_cfg,
                                         act_cfg=dict(type="ReLU"), inplace=False)
        elif rc_fusion == "cross_attention":
            self.cross_attention = Cross_Modal_Fusion(kernel_size=3, norm_cfg=norm_cfg)
        use_bev_conv(self)
        self.freeze_img = freeze_img
        self.freeze()

    def extract_feat(self, points, img, img_metas, gt_bboxes_3d=None):
        if self.rc_fusion != "cross_attention":
            return super().extract_feat(points, img, img_metas, gt_bboxes_3d)
        vox = self.voxelize_begin(points) if self.with_pts_backbone and points is not None else None
        img_feats = self.extract_img_feat(img, img_metas)
        pts_feats = self.extract_pts_feat(points, img_feats, img_metas,
                                          voxelized=None if vox is None else self.voxelize_end(vox))
        depth_dist = None
        if self.lift:
            BN, C, H, W = img_feats[0].shape
            view = img_feats[0].view(BN // self.num_views, self.num_views, C, H, W)
            rots, trans = self._cam_inverse(img_metas, view.device)
            img_bev_feat, depth_dist = self.lift_splat_shot_vis(view, rots, trans, lidar2img_rt=None, img_metas=img_metas)
            if pts_feats is None:
                pts_feats = [img_bev_feat]
            else:
                if img_bev_feat.shape[2:] != pts_feats[0].shape[2:]:
                    img_bev_feat = F.interpolate(img_bev_feat, pts_feats[0].shape[2:], mode="bilinear", align_corners=True)
                pts_feats = [self.cross_attention(img_bev_feat, pts_feats[0])]
        return dict(img_feats=img_feats, pts_feats=pts_feats, depth_dist=depth_dist)

"""Multi-task fusion detector (occupancy and/or 3-D boxes from the fused camera + radar BEV feature), SURVEY.md
8(f) rank 4 — mirror of the reference's ``BEVF_FasterRCNN_MTL``
(projects/mmdet3d_plugin/bevfusion/detectors/bevf_faster_rcnn_MTL.py:32-326).  Everything up to the fused BEV
feature is the BEVFUSION_depth path (same kernels); differences kept from the reference: the fusion conv maps
lic+imc -> imc channels and the SE block has imc channels (:62-74), targets travel as one dict, the head is a
``MultiTaskHeadv2``."""
import torch

from omnihd_amd.mm import DETECTORS, ConvModule
from omnihd_amd.mm.boxes import bbox3d2result
from omnihd_amd.mm.bricks import use_bev_conv
from omnihd_amd.mm.detector import MVXFasterRCNN

from ..dense_heads import mtl_occ_det_headv2, bev_occ_head  # noqa: F401  (register the heads)
from .bevf_faster_rcnn_bevdepth import BEVFUSION_depth, SE_Block
from .cam_stream_lss_bevpoolv2_depthnet import LiftSplatShoot_Depth


@DETECTORS.register_module()
class BEVF_FasterRCNN_MTL(BEVFUSION_depth):
    def __init__(self, freeze_img=False, lss=False, lc_fusion=False, camera_stream=False,
                 camera_depth_range=[4.0, 45.0, 1.0], img_depth_loss_weight=1.0, img_depth_loss_method="kld",
                 grid=0.6, num_views=6, se=False, final_dim=(900, 1600), pc_range=[-50, -50, -5, 50, 50, 3],
                 downsample=4, imc=256, lic=384, use_semantic=True, norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01),
                 **kwargs):
        MVXFasterRCNN.__init__(self, **kwargs)
        self.num_views, self.lc_fusion = num_views, lc_fusion
        self.img_depth_loss_weight, self.img_depth_loss_method = img_depth_loss_weight, img_depth_loss_method
        self.camera_depth_range = camera_depth_range
        self.lift, self.se, self.use_semantic = camera_stream, se, use_semantic
        if camera_stream:
            self.lift_splat_shot_vis = LiftSplatShoot_Depth(lss=lss, grid=grid, inputC=imc, camC=64, pc_range=pc_range,
                                                            camera_depth_range=camera_depth_range, final_dim=final_dim,
                                                            downsample=downsample, norm_cfg=norm_cfg)
        if lc_fusion:
            if se:
                self.seblock = SE_Block(imc)
            self.reduc_conv = ConvModule(lic + imc, imc, 3, padding=1, conv_cfg=None, norm_cfg=norm_cfg,
                                         act_cfg=dict(type="ReLU"), inplace=False)
        use_bev_conv(self)
        self.freeze_img = freeze_img
        self.freeze()

    def forward_pts_train(self, pts_feats, img_metas, mtl_targets):
        outs = self.pts_bbox_head(pts_feats, targets=mtl_targets)
        return self.pts_bbox_head.loss(predictions=outs, img_metas=img_metas, targets=mtl_targets)

    def forward_train(self, points=None, img_metas=None, gt_occ=None, gt_bboxes_3d=None, gt_labels_3d=None, gt_labels=None,
                      gt_bboxes=None, img=None, img_depth=None, proposals=None, gt_bboxes_ignore=None):
        fd = self.extract_feat(points, img=img, img_metas=img_metas, gt_bboxes_3d=gt_bboxes_3d)
        img_feats, pts_feats, depth_dist = fd["img_feats"], fd["pts_feats"], fd["depth_dist"]
        targets = dict(gt_bboxes_3d=gt_bboxes_3d, gt_labels_3d=gt_labels_3d, gt_bboxes_ignore=gt_bboxes_ignore, gt_occ=gt_occ)
        losses = dict()
        if pts_feats:
            losses.update(self.forward_pts_train(pts_feats, img_metas, targets))
        if img_feats:
            if img_depth is not None:
                loss_depth, _ = self.lift_splat_shot_vis.get_depth_loss(depth_labels=img_depth, depth_preds=depth_dist,
                                                                        loss_depth_type=self.img_depth_loss_method)
                losses.update(img_depth_loss=self.img_depth_loss_weight * loss_depth)
            losses.update(self.forward_img_train(img_feats, img_metas=img_metas))
        return losses

    def simple_test_pts(self, x, img_metas, rescale=False):
        outs = self.pts_bbox_head(x)
        predictions = self.pts_bbox_head.inference(outs, img_metas, rescale=rescale)
        if "bbox_list" in predictions:
            predictions["bbox_results"] = [bbox3d2result(b, s, l) for b, s, l in predictions.pop("bbox_list")]
        return predictions

    @torch.no_grad()
    def simple_test(self, points, img_metas, img=None, gt_occ=None, rescale=False):
        """-> {'bbox_results': [...], 'occ_pred': (B, Dx, Dy, Dz) class map, 'occ_results': (B, n_cls, 3) counters
        when ``gt_occ`` is given (reference :217-226)}."""
        fd = self.extract_feat(points, img=img, img_metas=img_metas)
        predictions = self.simple_test_pts(fd["pts_feats"], img_metas, rescale=rescale)
        if "bbox_results" in predictions:
            predictions["bbox_results"] = [dict(pts_bbox=r) for r in predictions["bbox_results"]]
        if "occ_pred" in predictions:
            occ = predictions.pop("occ_pred")
            predictions["occ_pred"] = (occ.softmax(-1).argmax(-1) if self.use_semantic
                                       else torch.sigmoid(occ[..., 0]))
            if gt_occ is not None and self.use_semantic:            # reference :217-222: score inside the model
                from ...datasets.evaluation_metrics import aug_evaluation_semantic
                gt = gt_occ[0] if isinstance(gt_occ, (list, tuple)) else gt_occ
                predictions["occ_results"] = aug_evaluation_semantic(predictions["occ_pred"], gt, img_metas[0],
                                                                     occ.shape[-1])
        return predictions

    def forward_test(self, points=None, img_metas=None, img=None, gt_occ=None, **kwargs):
        for var, name in [(points, "points"), (img_metas, "img_metas")]:
            if not isinstance(var, list):
                raise TypeError(f"{name} must be a list, but got {type(var)}")
        if len(points) != len(img_metas):
            raise ValueError(f"num of augmentations ({len(points)}) != num of image meta ({len(img_metas)})")
        if len(points) != 1:
            raise NotImplementedError("test-time augmentation is not part of the reference configs")
        return self.simple_test(points[0], img_metas[0], None if img is None else img[0], gt_occ, **kwargs)

"""``MVXFasterRCNN`` skeleton (mmdet3d v0.17.1 MVXTwoStageDetector subset) — the base class of the
reference's fusion detectors (bevfusion/detectors/bevf_faster_rcnn_bevdepth.py:33).  Only what the
camera + radar BEV-fusion path uses is restated: sub-module construction from the config dict with
the upstream attribute names (``pts_voxel_layer``, ``pts_voxel_encoder``, ``pts_middle_encoder``,
``pts_backbone``, ``pts_neck``, ``img_backbone``, ``img_neck``, ``pts_bbox_head``), ``voxelize``,
``extract_img_feat`` and ``forward_pts_train``.

The class is also a detector in its own right, registered under its upstream name: the reference's
radar-only and LiDAR-only PointPillars configs (projects/configs/bevfusion_NewScenes/radar_stream/
pointpillars_4DRadar.py:23 — the stage-1 run whose checkpoint the fusion config loads with
``load_from`` — RCFusion_NewScenes/radar_stream/RadarPillarNet.py and PointPillars_NewScenes/*.py)
use ``type='MVXFasterRCNN'`` directly, i.e. upstream's ``extract_pts_feat`` / ``extract_feat`` /
``forward_train`` / ``simple_test``.  That is the radar half of the hot path on its own: HIP hard
voxelisation -> pillar feature net -> HIP pillar scatter -> SECOND -> SECONDFPN -> anchor head."""
import torch
import torch.nn.functional as F
from torch import nn

from .registry import BACKBONES, DETECTORS, HEADS, MIDDLE_ENCODERS, NECKS, VOXEL_ENCODERS
from .boxes import bbox3d2result
from .second import Voxelization
from . import anchor_head, fpn, hard_vfe, resnet, second  # noqa: F401  (register upstream type names)


@DETECTORS.register_module()
class MVXFasterRCNN(nn.Module):
    def __init__(self, pts_voxel_layer=None, pts_voxel_encoder=None, pts_middle_encoder=None, pts_fusion_layer=None,
                 img_backbone=None, pts_backbone=None, img_neck=None, pts_neck=None, pts_bbox_head=None,
                 img_roi_head=None, img_rpn_head=None, train_cfg=None, test_cfg=None, pretrained=None, **_):
        super().__init__()
        if pts_voxel_layer:
            self.pts_voxel_layer = Voxelization(**pts_voxel_layer)
        if pts_voxel_encoder:
            self.pts_voxel_encoder = VOXEL_ENCODERS.build(pts_voxel_encoder)
        if pts_middle_encoder:
            self.pts_middle_encoder = MIDDLE_ENCODERS.build(pts_middle_encoder)
        if pts_backbone:
            self.pts_backbone = BACKBONES.build(pts_backbone)
        if pts_neck is not None:
            self.pts_neck = NECKS.build(pts_neck)
        if pts_bbox_head:
            head = dict(pts_bbox_head)
            head.update(train_cfg=train_cfg["pts"] if train_cfg else None, test_cfg=test_cfg["pts"] if test_cfg else None)
            self.pts_bbox_head = HEADS.build(head)
        if img_backbone:
            self.img_backbone = BACKBONES.build(img_backbone)
        if img_neck is not None:
            self.img_neck = NECKS.build(img_neck)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg

    with_pts_backbone = property(lambda self: hasattr(self, "pts_backbone"))
    with_pts_neck = property(lambda self: hasattr(self, "pts_neck"))
    with_img_backbone = property(lambda self: hasattr(self, "img_backbone"))
    with_img_neck = property(lambda self: hasattr(self, "img_neck"))
    with_pts_bbox = property(lambda self: hasattr(self, "pts_bbox_head"))
    with_img_bbox = property(lambda self: False)

    def extract_img_feat(self, img, img_metas):
        """(B, N, C, H, W) images -> list of (B*N, C', h, w) neck outputs."""
        if not self.with_img_backbone or img is None:
            return None
        if img.dim() == 5:
            B, N, C, H, W = img.shape
            img = img.reshape(B * N, C, H, W)
        feats = self.img_backbone(img)
        if self.with_img_neck:
            feats = self.img_neck(feats)
        return feats

    @torch.no_grad()
    def voxelize_begin(self, points):
        """Enqueue the per-sample hard voxelisation (HIP kernels); nothing is read back yet."""
        return [self.pts_voxel_layer.begin(res) for res in points]

    @torch.no_grad()
    def voxelize_end(self, pending):
        """Voxel counts arrive (one event wait per sample), outputs are concatenated, batch index prepended."""
        voxels, coors, nums = [], [], []
        for i, h in enumerate(pending):
            v, c, n = h.get()
            voxels.append(v)
            nums.append(n)
            coors.append(F.pad(c, (1, 0), mode="constant", value=i))
        return torch.cat(voxels, dim=0), torch.cat(nums, dim=0), torch.cat(coors, dim=0)

    def voxelize(self, points):
        return self.voxelize_end(self.voxelize_begin(points))

    # ---- upstream MVXTwoStageDetector bodies (subclasses of the reference override most of them) ----
    def extract_pts_feat(self, pts, img_feats, img_metas):
        if not self.with_pts_bbox:
            return None
        voxels, num_points, coors = self.voxelize(pts)
        voxel_features = self.pts_voxel_encoder(voxels, num_points, coors, img_feats, img_metas, **self._encoder_hints(pts))
        # upstream reads the batch size back from the device (coors[-1, 0] + 1); the host already knows it
        x = self.pts_middle_encoder(voxel_features, coors, len(pts))
        x = self.pts_backbone(x)
        if self.with_pts_neck:
            x = self.pts_neck(x)
        return x

    def _encoder_hints(self, pts):
        """Host-known facts a voxel encoder can use without asking the device: HardVFE's packed evaluation sizes its
        point buffer by the number of input points (an upper bound of the occupied slots)."""
        from .hard_vfe import HardVFE
        if isinstance(getattr(self, "pts_voxel_encoder", None), HardVFE):
            return dict(max_real_points=sum(int(p.shape[0]) for p in pts))
        return {}

    def extract_feat(self, points, img, img_metas):
        img_feats = self.extract_img_feat(img, img_metas)
        return img_feats, self.extract_pts_feat(points, img_feats, img_metas)

    def forward_train(self, points=None, img_metas=None, gt_bboxes_3d=None, gt_labels_3d=None, gt_labels=None,
                      gt_bboxes=None, img=None, proposals=None, gt_bboxes_ignore=None):
        img_feats, pts_feats = self.extract_feat(points, img=img, img_metas=img_metas)
        losses = dict()
        if pts_feats:
            losses.update(self.forward_pts_train(pts_feats, gt_bboxes_3d, gt_labels_3d, img_metas, gt_bboxes_ignore))
        if img_feats:
            losses.update(self.forward_img_train(img_feats, img_metas=img_metas))
        return losses

    @torch.no_grad()
    def simple_test(self, points, img_metas, img=None, rescale=False):
        _, pts_feats = self.extract_feat(points, img=img, img_metas=img_metas)
        bbox_list = [dict() for _ in range(len(img_metas))]
        if pts_feats and self.with_pts_bbox:
            for result, pts_bbox in zip(bbox_list, self.simple_test_pts(pts_feats, img_metas, rescale=rescale)):
                result["pts_bbox"] = pts_bbox
        return bbox_list

    def forward(self, return_loss=True, **kwargs):
        return self.forward_train(**kwargs) if return_loss else self.forward_test(**kwargs)

    def forward_pts_train(self, pts_feats, gt_bboxes_3d, gt_labels_3d, img_metas, gt_bboxes_ignore=None):
        outs = self.pts_bbox_head(pts_feats)
        return self.pts_bbox_head.loss(*outs, gt_bboxes_3d, gt_labels_3d, img_metas, gt_bboxes_ignore=gt_bboxes_ignore)

    def forward_img_train(self, x, img_metas, **kwargs):
        return dict()     # no image head in the fusion configs

    # ---- test time (mmdet3d MVXTwoStageDetector.simple_test_pts / forward_test / Base3DDetector.forward) ----
    def simple_test_pts(self, x, img_metas, rescale=False):
        outs = self.pts_bbox_head(x)
        bbox_list = self.pts_bbox_head.get_bboxes(*outs, img_metas, rescale=rescale)
        return [bbox3d2result(b, s, l) for b, s, l in bbox_list]

    def forward_test(self, points=None, img_metas=None, img=None, **kwargs):
        """Upstream wraps every test input in a list of augmentations; the configs use one (no TTA)."""
        for var, name in [(points, "points"), (img_metas, "img_metas")]:
            if not isinstance(var, list):
                raise TypeError(f"{name} must be a list, but got {type(var)}")
        if len(points) != len(img_metas):
            raise ValueError(f"num of augmentations ({len(points)}) != num of image meta ({len(img_metas)})")
        if len(points) != 1:
            raise NotImplementedError("test-time augmentation is not part of the reference configs")
        return self.simple_test(points[0], img_metas[0], None if img is None else img[0], **kwargs)

"""``BEVFusionTripleTemporal`` — camera + 4D-radar + LiDAR BEV fusion over a queue of frames:
BASELINE.json's last configuration ("cam+radar+LiDAR triple-modal fusion + 4-frame temporal BEV").

THE REFERENCE HAS NO COUNTERPART (SURVEY.md defect D11): every NewScenes fusion config sets
``use_lidar=False`` and temporal queues exist only in its BEVFormer branch.  This file is therefore a
COMPOSITION of reference ingredients, and its parity is unpinned by construction — what is checked is
that every ingredient is the already-pinned one and that the HIP path agrees with the oracle path:

* camera and radar streams, fusion conv + SE, head, losses: ``BEVFUSION_depth`` unchanged
  (bevf_faster_rcnn_bevdepth.py:33-232);
* LiDAR stream: the modules of projects/configs/PointPillars_NewScenes/pointpillars_LiDAR.py:23-52
  (hard voxelisation with 64 points per pillar, upstream ``HardVFE``, ``PointPillarsScatter``,
  ``SECOND``, ``SECONDFPN``) held as a head-less ``MVXFasterRCNN`` under ``lidar_stream`` so that a
  checkpoint of that config loads into it by prefix; the three BEV maps are concatenated in front of the
  same ``reduc_conv`` (imc + 2*lic -> lic);
* temporal queue: the recipe of the reference's BEVFormer detector (bevformer/detectors/bevformer.py:
  ``obtain_history_bev``): the T-1 history frames run in eval mode under ``no_grad``, only the last
  frame carries gradients.  Here the history frames of the whole batch go through the network as ONE
  flat batch of B*(T-1) samples (large launches instead of T-1 small ones), their fused BEV maps are
  resampled into the current ego frame (``ego_delta`` = (dx, dy, dyaw) of the frame's ego pose in the
  current LiDAR frame; bilinear, zeros outside) and a 3x3 ``temporal_conv`` (T*lic -> lic) mixes the
  queue.

Inputs (queue-major, as BEVFormer's): ``img (B, T, N, 3, H, W)``; ``points`` / ``lidar_points`` = list
over B of lists over T of (n, C) tensors; ``img_metas`` = list over B of lists over T of dicts
(``lidar2img``, optional ``ego_delta`` and ``history_valid`` — ``datasets/temporal_queue.py`` derives both from the
reference's ``can_bus`` / ``scene_token`` metas); ground truth and ``img_depth (B, N, H, W)`` belong to the last
frame.  No CPU path: the HIP operators raise on CPU tensors.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from omnihd_amd.mm import DETECTORS, ConvModule
from omnihd_amd.mm.bricks import use_bev_conv
from omnihd_amd.mm.detector import MVXFasterRCNN

from .bevf_faster_rcnn_bevdepth import BEVFUSION_depth

__all__ = ["BEVFusionTripleTemporal", "bev_warp_theta"]


def bev_warp_theta(ego_delta, pc_range):
    """2x3 ``affine_grid`` matrix that resamples a history BEV map into the current ego frame.

    ``ego_delta = (dx, dy, dyaw)``: a point p_hist of the history frame sits at R(dyaw) p_hist + (dx, dy) in
    the current frame.  The output cell at current position p therefore reads the history map at
    R(dyaw)^T (p - t).  Normalised grid coordinates are x_n = (x - cx) / hx with (cx, hx) the centre and
    half-extent of ``pc_range`` along x (and likewise y)."""
    dx, dy, yaw = (float(v) for v in ego_delta)
    c, s = math.cos(yaw), math.sin(yaw)
    cx, cy = (pc_range[0] + pc_range[3]) / 2.0, (pc_range[1] + pc_range[4]) / 2.0
    hx, hy = (pc_range[3] - pc_range[0]) / 2.0, (pc_range[4] - pc_range[1]) / 2.0
    to_metric = np.array([[hx, 0, cx], [0, hy, cy], [0, 0, 1.0]])
    inv_motion = np.array([[c, s, -(c * dx + s * dy)], [-s, c, -(-s * dx + c * dy)], [0, 0, 1.0]])
    return (np.linalg.inv(to_metric) @ inv_motion @ to_metric)[:2]


@DETECTORS.register_module()
class BEVFusionTripleTemporal(BEVFUSION_depth):
    def __init__(self, lidar_stream=None, queue_length=4, imc=256, lic=384, lc_fusion=True,
                 norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), **kwargs):
        if not lidar_stream or not lc_fusion:
            raise ValueError("BEVFusionTripleTemporal needs lidar_stream=dict(pts_voxel_layer=..., ...) and lc_fusion=True")
        super().__init__(imc=imc, lic=lic, lc_fusion=True, norm_cfg=norm_cfg, **kwargs)
        if not (self.lift and self.with_pts_backbone):
            raise ValueError("BEVFusionTripleTemporal needs the camera stream and the radar stream as well")
        self.queue_length = int(queue_length)
        keep = ("pts_voxel_layer", "pts_voxel_encoder", "pts_middle_encoder", "pts_backbone", "pts_neck")
        self.lidar_stream = MVXFasterRCNN(**{k: lidar_stream[k] for k in keep})
        # the fusion conv sees the third BEV map as well; the queue is mixed by one more 3x3 conv block
        self.reduc_conv = ConvModule(imc + 2 * lic, lic, 3, padding=1, conv_cfg=None, norm_cfg=norm_cfg,
                                     act_cfg=dict(type="ReLU"), inplace=False)
        self.temporal_conv = ConvModule(self.queue_length * lic, lic, 3, padding=1, conv_cfg=None, norm_cfg=norm_cfg,
                                        act_cfg=dict(type="ReLU"), inplace=False)
        self._pc_range = list(kwargs.get("pc_range", [-50, -50, -5, 50, 50, 3]))
        use_bev_conv(self)
        self.freeze()

    # ---- one time step, flat batch ------------------------------------------------------------------------
    def _lidar_feat(self, pending, lidar_points):
        s = self.lidar_stream
        voxels, num_points, coors = s.voxelize_end(pending)
        feats = s.pts_voxel_encoder(voxels, num_points, coors, **s._encoder_hints(lidar_points))
        x = s.pts_middle_encoder(feats, coors, len(lidar_points))
        return s.pts_neck(s.pts_backbone(x))

    def extract_feat(self, points, img, img_metas, lidar_points=None, gt_bboxes_3d=None):
        """Fused BEV map (B', lic, Y, X) of one time step for a flat batch of B' samples."""
        if lidar_points is None:
            raise ValueError("lidar_points is required")
        radar_vox = self.voxelize_begin(points)                   # both voxelisations are enqueued first,
        lidar_vox = self.lidar_stream.voxelize_begin(lidar_points)   # their counts are awaited after the image branch
        img_feats = self.extract_img_feat(img, img_metas)
        BN, C, H, W = img_feats[0].shape
        view = img_feats[0].view(BN // self.num_views, self.num_views, C, H, W)
        rots, trans = self._cam_inverse(img_metas, view.device)
        cam_bev, depth_dist = self.lift_splat_shot_vis(view, rots, trans, lidar2img_rt=None, img_metas=img_metas)
        radar_bev = self.extract_pts_feat(points, img_feats, img_metas, voxelized=self.voxelize_end(radar_vox))[0]
        lidar_bev = self._lidar_feat(lidar_vox, lidar_points)[0]
        if cam_bev.shape[2:] != radar_bev.shape[2:]:
            cam_bev = F.interpolate(cam_bev, radar_bev.shape[2:], mode="bilinear", align_corners=True)
        fused = self.reduc_conv(torch.cat([cam_bev, radar_bev, lidar_bev], dim=1))
        if self.se:
            fused = self.seblock(fused)
        return dict(img_feats=img_feats, pts_feats=[fused], depth_dist=depth_dist)

    # ---- the queue ------------------------------------------------------------------------------------------
    def _check_queue(self, img, points, lidar_points, img_metas):
        B, T = img.shape[:2]
        if T != self.queue_length:
            raise ValueError(f"queue of {T} frames, the model was built for {self.queue_length}")
        for name, seq in (("points", points), ("lidar_points", lidar_points), ("img_metas", img_metas)):
            if len(seq) != B or any(len(s) != T for s in seq):
                raise ValueError(f"{name} must be a list over {B} samples of lists over {T} frames")
        return B, T

    def _history_bev(self, points, lidar_points, img, img_metas):
        """Fused BEV maps of frames 0..T-2 resampled into the frame T-1, (B, (T-1)*lic, Y, X); no gradients."""
        B, T = img.shape[:2]
        flat = lambda seq: [seq[b][t] for b in range(B) for t in range(T - 1)]      # noqa: E731
        was_training = self.training
        self.eval()
        try:
            with torch.no_grad():
                bev = self.extract_feat(flat(points), img[:, :T - 1].flatten(0, 1), flat(img_metas),
                                        lidar_points=flat(lidar_points))["pts_feats"][0]
                metas = flat(img_metas)
                if any("ego_delta" in m for m in metas):
                    theta = np.stack([bev_warp_theta(m.get("ego_delta", (0.0, 0.0, 0.0)), self._pc_range) for m in metas])
                    # sampling positions in fp32 whatever the autocast dtype: a bf16 grid is off by whole cells
                    theta = torch.from_numpy(theta).to(device=bev.device, dtype=torch.float32)
                    with torch.autocast(bev.device.type, enabled=False):
                        grid = F.affine_grid(theta, list(bev.shape), align_corners=False)
                        bev = F.grid_sample(bev.float(), grid, mode="bilinear", padding_mode="zeros",
                                            align_corners=False).to(bev.dtype)
                if not all(m.get("history_valid", True) for m in metas):      # frames of another scene contribute nothing
                    keep = torch.tensor([bool(m.get("history_valid", True)) for m in metas], device=bev.device)
                    bev = bev * keep.view(-1, 1, 1, 1).to(bev.dtype)
        finally:
            self.train(was_training)
        return bev.reshape(B, (T - 1) * bev.shape[1], *bev.shape[2:])

    def extract_queue_feat(self, points, lidar_points, img, img_metas):
        B, T = self._check_queue(img, points, lidar_points, img_metas)
        last = lambda seq: [s[T - 1] for s in seq]                                     # noqa: E731
        hist = self._history_bev(points, lidar_points, img, img_metas) if T > 1 else None
        fd = self.extract_feat(last(points), img[:, T - 1], last(img_metas), lidar_points=last(lidar_points))
        cur = fd["pts_feats"][0]
        x = cur if hist is None else torch.cat([hist.to(cur.dtype), cur], dim=1)
        fd["pts_feats"] = [self.temporal_conv(x)]
        return fd

    def forward_train(self, points=None, img_metas=None, gt_bboxes_3d=None, gt_labels_3d=None, gt_labels=None,
                      gt_bboxes=None, img=None, img_depth=None, proposals=None, gt_bboxes_ignore=None, lidar_points=None):
        fd = self.extract_queue_feat(points, lidar_points, img, img_metas)
        cur_metas = [m[-1] for m in img_metas]
        losses = dict(self.forward_pts_train(fd["pts_feats"], gt_bboxes_3d, gt_labels_3d, cur_metas, gt_bboxes_ignore))
        if img_depth is not None:
            losses.update(img_depth_loss=self.img_depth_loss_weight * self._depth_loss(fd["depth_dist"], img_depth))
        return losses

    @torch.no_grad()
    def simple_test(self, points, img_metas, img=None, rescale=False, lidar_points=None):
        fd = self.extract_queue_feat(points, lidar_points, img, img_metas)
        cur_metas = [m[-1] for m in img_metas]
        return [dict(pts_bbox=r) for r in self.simple_test_pts(fd["pts_feats"], cur_metas, rescale=rescale)]

    def forward_test(self, points=None, img_metas=None, img=None, lidar_points=None, **kwargs):
        for var, name in [(points, "points"), (img_metas, "img_metas"), (lidar_points, "lidar_points")]:
            if not isinstance(var, list) or len(var) != 1:
                raise TypeError(f"{name} must be a list holding one (un-augmented) entry")
        return self.simple_test(points[0], img_metas[0], img[0], lidar_points=lidar_points[0], **kwargs)

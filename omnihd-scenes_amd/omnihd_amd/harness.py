"""Synthetic-input training-step harness for the camera + 4D-radar BEV-fusion detector.

Builds the detector from the REFERENCE's config dict (projects/configs/bevfusion_NewScenes/
bevfusion.py, loaded unchanged when the file is available, otherwise from the equivalent dict kept
here) and runs forward + backward + AdamW on seeded synthetic inputs of the shapes SURVEY.md 8(d)
specifies.  The reference drives the same step through mmcv's EpochBasedRunner
(projects/mmdet3d_plugin/bevformer/apis/mmdet_train.py:76-207: MMDistributedDataParallel with
broadcast_buffers=False, AdamW lr 2e-4 wd 0.05, grad-clip 35); the runner itself is out of scope.
"""
import copy
import math
import os

import numpy as np
import torch
from torch import nn

POINT_CLOUD_RANGE = [-60.0, -40.0, -3.0, 60.0, 40.0, 5.0]
RES = {"r1": (256, 704, 410.0), "r2": (544, 960, 560.0)}
ANCHOR_SIZES = [[1.9768212501227105, 4.637021209998035, 1.6647611354273741],
                [0.796163784946599, 0.8183815295280997, 1.6895737765415433],
                [0.912318683145357, 1.9201067650572057, 1.620921669034068],
                [2.6724696700336494, 8.184714524976142, 3.0254503871391982]]
ANCHOR_Z = [0.9104247242165809, 1.1421614665993767, 0.9059764319390522, 1.5158325603046292]


def seed_miopen_db():
    """MIOpen tuning records for the convolution geometries of the benchmarked / tested workloads (find-db, perf-db and the
    compiled kernels of the chosen solvers, captured on an MI355X with this image; ``omnihd-scenes_amd/miopen_db``): copied to
    a scratch directory and offered to MIOpen as its user database, so that the find step and the immediate-mode kernel
    builds of a fresh box are lookups (bench start-up 3m52s -> 1m52s; the first step of the bs=2 four-frame configuration
    457 s -> under two minutes).  Must run before the first convolution of the process; a missing or rejected database only
    costs the time back.  Returns the directory in use (or None)."""
    import shutil
    import tempfile
    if "MIOPEN_USER_DB_PATH" in os.environ:
        return os.environ["MIOPEN_USER_DB_PATH"]
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "miopen_db")
    if not os.path.isdir(src):
        return None
    dst = os.path.join(tempfile.gettempdir(), "omnihd_miopen_%d_%s" % (os.getuid(), os.environ.get("LOCAL_RANK", "0")))
    try:
        os.makedirs(dst, exist_ok=True)
        for name in os.listdir(src):
            if not os.path.exists(os.path.join(dst, name)):
                shutil.copy(os.path.join(src, name), os.path.join(dst, name))
        os.environ["MIOPEN_USER_DB_PATH"] = dst
        os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = dst
        return dst
    except OSError:
        return None


def reference_model_cfg():
    """The ``model=dict(...)`` of projects/configs/bevfusion_NewScenes/bevfusion.py:30-155, restated as
    data (used when the reference checkout is not on the machine, e.g. the GPU box)."""
    pcr, vs = POINT_CLOUD_RANGE, [0.25, 0.25, 8]
    sync1, sync2 = (dict(type="naiveSyncBN1d", eps=1e-3, momentum=0.01), dict(type="naiveSyncBN2d", eps=1e-3, momentum=0.01))
    return dict(
        type="BEVFUSION_depth", freeze_img=False, se=True, lc_fusion=True, camera_stream=True, lss=False, grid=0.5,
        num_views=6, final_dim=(544, 960), pc_range=pcr, downsample=4, camera_depth_range=[1, 60, 1],
        img_depth_loss_method="kld", img_depth_loss_weight=1.0,
        pts_voxel_layer=dict(max_num_points=10, point_cloud_range=pcr, voxel_size=vs, max_voxels=(30000, 40000)),
        pts_voxel_encoder=dict(type="PillarFeatureNetV1", in_channels=8, feat_channels=[64], with_distance=False,
                               voxel_size=vs, point_cloud_range=pcr, norm_cfg=sync1),
        pts_middle_encoder=dict(type="PointPillarsScatter", in_channels=64, output_shape=[320, 480]),
        pts_backbone=dict(type="SECOND", in_channels=64, norm_cfg=sync2, layer_nums=[3, 5, 5], layer_strides=[2, 2, 2],
                          out_channels=[64, 128, 256]),
        pts_neck=dict(type="SECONDFPN", norm_cfg=sync2, in_channels=[64, 128, 256], upsample_strides=[1, 2, 4],
                      out_channels=[128, 128, 128]),
        img_backbone=dict(type="ResNet", depth=50, num_stages=4, out_indices=(1, 2, 3), frozen_stages=1,
                          norm_cfg=dict(type="BN", requires_grad=False), norm_eval=True, style="pytorch"),
        img_neck=dict(type="FPNC", final_dim=(544, 960), downsample=4, in_channels=[512, 1024, 2048], out_channels=256,
                      use_adp=True, num_outs=4),
        pts_bbox_head=dict(
            type="Anchor3DHead", num_classes=4, in_channels=384, feat_channels=384, use_direction_classifier=True,
            anchor_generator=dict(type="AlignedAnchor3DRangeGenerator",
                                  ranges=[[-60, -40, z, 60, 40, z] for z in ANCHOR_Z], sizes=ANCHOR_SIZES,
                                  custom_values=[0, 0], rotations=[0, 1.57], reshape_out=True),
            assigner_per_size=False, diff_rad_by_sin=True, dir_offset=0.7854, dir_limit_offset=0,
            bbox_coder=dict(type="DeltaXYZWLHRBBoxCoder", code_size=9),
            loss_cls=dict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
            loss_bbox=dict(type="SmoothL1Loss", beta=1.0 / 9.0, loss_weight=1.0),
            loss_dir=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=0.2)),
        train_cfg=dict(pts=dict(assigner=dict(type="MaxIoUAssigner", iou_calculator=dict(type="BboxOverlapsNearest3D"),
                                              pos_iou_thr=0.6, neg_iou_thr=0.3, min_pos_iou=0.3, ignore_iof_thr=-1),
                                allowed_border=0, code_weight=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2],
                                pos_weight=-1, debug=False)),
        test_cfg=dict(pts=dict(use_rotate_nms=True, nms_across_levels=False, nms_pre=1000, nms_thr=0.2, score_thr=0.05,
                               min_bbox_size=0, max_num=500)))


def model_cfg_for(res, radar_dims):
    """Reference config with only the input geometry adapted: final_dim (R1 = the BASELINE metric's
    256x704; R2 = the repo's 544x960) and the number of radar channels (7 in BASELINE.json, 8 in the
    repo config)."""
    ref = "/root/reference/projects/configs/bevfusion_NewScenes/bevfusion.py"
    if os.path.exists(ref):
        from .mm.config import load_config
        cfg = copy.deepcopy(load_config(ref)["model"])
    else:
        cfg = reference_model_cfg()
    H, W, _ = RES[res]
    cfg["final_dim"] = (H, W)
    cfg["img_neck"]["final_dim"] = (H, W)
    cfg["pts_voxel_encoder"]["in_channels"] = radar_dims
    return cfg


TINY = dict(H=32, W=48, fx=30.0, pc_range=[-8.0, -6.0, -1.0, 8.0, 6.0, 1.0], grid=1.0, voxel_size=[0.5, 0.5, 2.0],
            canvas=[24, 32], depth_range=[1.0, 9.0, 1.0], radius=0.3, height=0.2, n_points=(300, 600), n_boxes=6,
            anchor_sizes=[[0.8, 1.6, 0.8], [0.4, 0.4, 0.9]], anchor_z=[-0.4, -0.45])


def tiny_model_cfg(radar_dims=7):
    """A scaled-down copy of the reference model config (same module graph: R50, FPNC, DepthNet with
    DCN, LSS, pillars, SECOND/FPN, fusion conv + SE, Anchor3DHead) for tests and smoke runs."""
    cfg = reference_model_cfg()
    t = TINY
    cfg.update(final_dim=(t["H"], t["W"]), pc_range=t["pc_range"], grid=t["grid"], camera_depth_range=t["depth_range"])
    cfg["img_neck"]["final_dim"] = (t["H"], t["W"])
    cfg["pts_voxel_layer"].update(point_cloud_range=t["pc_range"], voxel_size=t["voxel_size"], max_voxels=(400, 500))
    cfg["pts_voxel_encoder"].update(in_channels=radar_dims, voxel_size=t["voxel_size"], point_cloud_range=t["pc_range"])
    cfg["pts_middle_encoder"]["output_shape"] = t["canvas"]
    r = t["pc_range"]
    cfg["pts_bbox_head"].update(num_classes=2)
    cfg["pts_bbox_head"]["anchor_generator"].update(
        ranges=[[r[0], r[1], z, r[3], r[4], z] for z in t["anchor_z"]], sizes=t["anchor_sizes"])
    return cfg


def occ_model_cfg(base):
    """``base`` (a BEVFUSION_depth config dict) turned into the multi-task occupancy config of
    projects/configs/bevfusion_NewScenes/bevfusion_occ.py:57-193: same trunk, detector BEVF_FasterRCNN_MTL, head
    MultiTaskHeadv2 with the occupancy task enabled (12 classes x 16 height bins) and detection disabled."""
    cfg = copy.deepcopy(base)
    det_head = cfg.pop("pts_bbox_head")
    det_head["type"] = "Anchor3DHeadV1"
    det_head.update(in_channels=256, feat_channels=256)
    r, g = cfg["pc_range"], cfg["grid"]
    grid = dict(xbound=[r[0], r[3], g], ybound=[r[1], r[4], g], zbound=[-10.0, 10.0, 20.0], dbound=list(cfg["camera_depth_range"]))
    cfg.update(type="BEVF_FasterRCNN_MTL", pts_bbox_head=dict(
        type="MultiTaskHeadv2", in_channels=256, out_channels=256, grid_conf=grid, det_grid_conf=grid, occ_grid_conf=grid,
        task_enbale={"3dod": False, "occ": True}, task_weights={"3dod": 1.0, "occ": 1.0}, bev_encode_block="Basic",
        cfg_3dod=det_head, cfg_occ=dict(type="BEVOCCHead2Dv2", in_dim=256, out_dim=256, num_classes=12, use_predicter=True,
                                        loss_occ=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=1.0))))
    return cfg


def camera_model_cfg(base):
    """``base`` (a BEVFUSION_depth config dict) cut down to the camera-only stage-1 config of the reference,
    projects/configs/bevfusion_NewScenes/cam_stream/LSS.py:30-123 (BASELINE.json configs[1]): no point stream, no fusion
    conv (``lc_fusion=False``), torch SyncBN with trainable affine parameters in the detector AND in the image backbone
    (``norm_eval=False``), detection head on the 256-channel camera BEV feature."""
    keep = ("type", "camera_stream", "lss", "grid", "num_views", "final_dim", "pc_range", "downsample", "camera_depth_range",
            "img_depth_loss_method", "img_depth_loss_weight", "img_backbone", "img_neck", "pts_bbox_head", "train_cfg", "test_cfg")
    cfg = {k: copy.deepcopy(base[k]) for k in keep}
    sync = dict(type="SyncBN", requires_grad=True)
    cfg.update(lc_fusion=False, norm_cfg=dict(sync))
    cfg["img_backbone"].update(norm_cfg=dict(sync), norm_eval=False)
    cfg["pts_bbox_head"].update(in_channels=256, feat_channels=256)
    return cfg


def pillars_model_cfg(base, stream="radar"):
    """``base`` (a BEVFUSION_depth config dict) cut down to the point-stream-only ``MVXFasterRCNN`` configs of the
    reference: ``stream='radar'`` = projects/configs/bevfusion_NewScenes/radar_stream/pointpillars_4DRadar.py:23-118
    (stage 1 of the fusion recipe; identical to PointPillars_NewScenes/pointpillars_4DRadar.py), ``'rcfusion'`` =
    RCFusion_NewScenes/radar_stream/RadarPillarNet.py (RadarPillarFeatureNet), ``'lidar'`` =
    PointPillars_NewScenes/pointpillars_LiDAR.py:21-118 (64 points per pillar, upstream HardVFE)."""
    keep = ("pts_voxel_layer", "pts_voxel_encoder", "pts_middle_encoder", "pts_backbone", "pts_neck", "pts_bbox_head",
            "train_cfg", "test_cfg")
    cfg = {k: copy.deepcopy(base[k]) for k in keep}
    cfg["type"] = "MVXFasterRCNN"
    enc = cfg["pts_voxel_encoder"]
    if stream == "rcfusion":
        enc.update(type="RadarPillarFeatureNet", in_channels=7, with_cluster_center=True, with_voxel_center=True,
                   with_velocity_snr_center=True)
    elif stream == "lidar":
        cfg["pts_voxel_layer"]["max_num_points"] = 64
        cfg["pts_voxel_encoder"] = dict(type="HardVFE", in_channels=4, feat_channels=[64, 64], with_distance=False,
                                        voxel_size=enc["voxel_size"], with_cluster_center=True, with_voxel_center=True,
                                        point_cloud_range=enc["point_cloud_range"], norm_cfg=enc["norm_cfg"])
    elif stream != "radar":
        raise ValueError(stream)
    return cfg


def triple_model_cfg(base, queue_length=4):
    """``base`` (a BEVFUSION_depth config dict) extended to BASELINE.json's last configuration: the LiDAR stream
    of pointpillars_LiDAR.py next to the radar stream, and a queue of ``queue_length`` frames (a
    composition, no reference config exists)."""
    cfg = copy.deepcopy(base)
    lidar = pillars_model_cfg(base, "lidar")
    cfg.update(type="BEVFusionTripleTemporal", queue_length=queue_length,
               lidar_stream={k: lidar[k] for k in ("pts_voxel_layer", "pts_voxel_encoder", "pts_middle_encoder",
                                                   "pts_backbone", "pts_neck")})
    return cfg


def synthetic_queue(res, batch, radar_dims, device, seed, frames=4, lidar_points=None):
    """Inputs of one step of the triple-modal temporal detector: ``frames`` time steps of images, radar and LiDAR
    clouds per sample (queue-major), ego motion of ~1 m / 1 degree per frame, ground truth for the last frame."""
    steps = [synthetic_batch(res, batch, radar_dims, device, seed + 17 * t) for t in range(frames)]
    rng = np.random.default_rng(seed + 5)
    r = TINY["pc_range"] if res == "tiny" else POINT_CLOUD_RANGE
    n_lidar = lidar_points or (2000 if res == "tiny" else 120000)
    lidar = []
    for _ in range(batch):
        seq = []
        for _ in range(frames):
            p = np.empty((n_lidar, 4), dtype=np.float32)
            # two thirds of the returns within the inner third of the range, as a spinning LiDAR's density falls off
            scale = np.where(rng.random(n_lidar) < 0.66, 0.33, 1.0)
            p[:, 0] = rng.uniform(r[0], r[3], n_lidar) * scale
            p[:, 1] = rng.uniform(r[1], r[4], n_lidar) * scale
            p[:, 2] = rng.uniform(r[2], r[5], n_lidar)
            p[:, 3] = rng.uniform(0, 1, n_lidar)
            seq.append(torch.from_numpy(p).to(device))
        lidar.append(seq)
    metas = []
    for b in range(batch):
        seq = []
        for t in range(frames):
            back = frames - 1 - t
            m = dict(steps[t]["img_metas"][b])
            m["ego_delta"] = (-1.0 * back, 0.05 * back, math.radians(-1.0 * back))
            seq.append(m)
        metas.append(seq)
    last = steps[-1]
    return dict(points=[[steps[t]["points"][b] for t in range(frames)] for b in range(batch)], lidar_points=lidar,
                img=torch.stack([s["img"] for s in steps], dim=1), img_metas=metas, img_depth=last["img_depth"],
                gt_bboxes_3d=last["gt_bboxes_3d"], gt_labels_3d=last["gt_labels_3d"])


def synthetic_occupancy(batch, nx, ny, nz, n_cls, device, seed):
    """(B, Dx, Dy, Dz) class map: free space (0) with boxes of the other classes, a few unknown (255) voxels are NOT
    generated — the reference's cross-entropy has no ignore index and would raise on them."""
    rng = np.random.default_rng(seed + 77)
    occ = np.zeros((batch, nx, ny, nz), dtype=np.int64)
    for b in range(batch):
        for _ in range(40):
            c = int(rng.integers(1, n_cls))
            x, y, z = int(rng.integers(0, nx)), int(rng.integers(0, ny)), int(rng.integers(0, nz))
            dx, dy, dz = (int(v) for v in rng.integers(1, 9, 3))
            occ[b, x:x + dx, y:y + dy, z:z + min(dz, 4)] = c
    return torch.from_numpy(occ).to(device)


def synthetic_lidar2img(res):
    """Six pinhole cameras on a ring: float64 4x4 lidar2img per camera."""
    if res == "tiny":
        H, W, fx, radius, height = TINY["H"], TINY["W"], TINY["fx"], TINY["radius"], TINY["height"]
    else:
        (H, W, fx), radius, height = RES[res], 1.0, 1.5
    mats = []
    for yaw_deg in (0, 60, -60, 180, 120, -120):
        yaw = math.radians(yaw_deg)
        R_c2l = np.array([[math.sin(yaw), 0, math.cos(yaw)], [-math.cos(yaw), 0, math.sin(yaw)], [0, -1, 0]])
        t_c2l = np.array([radius * math.cos(yaw), radius * math.sin(yaw), height])
        R = R_c2l.T
        E = np.eye(4); E[:3, :3] = R; E[:3, 3] = -R @ t_c2l
        K = np.eye(4); K[0, 0] = K[1, 1] = fx; K[0, 2] = W / 2; K[1, 2] = H / 2
        mats.append(K @ E)
    return mats


def synthetic_batch(res, batch, radar_dims, device, seed):
    """Inputs of one training step, resident on ``device``."""
    if res == "tiny":
        return _tiny_batch(batch, radar_dims, device, seed)
    H, W, _ = RES[res]
    g = torch.Generator(device="cpu").manual_seed(seed)
    rng = np.random.default_rng(seed)
    img = torch.randn(batch, 6, 3, H, W, generator=g)
    img_depth = torch.zeros(batch, 6, H, W)
    mask = torch.rand(batch, 6, H, W, generator=g) < 0.02
    img_depth[mask] = torch.rand(int(mask.sum()), generator=g) * 59 + 1
    points, gt_boxes, gt_labels = [], [], []
    for _ in range(batch):
        n = int(rng.integers(8000, 20001))
        p = np.empty((n, radar_dims), dtype=np.float32)
        p[:, 0] = rng.uniform(-60, 60, n); p[:, 1] = rng.uniform(-40, 40, n); p[:, 2] = rng.uniform(-3, 5, n)
        p[:, 3:5] = rng.normal(0, 5, (n, 2)); p[:, 5] = rng.uniform(0, 60, n); p[:, 6] = rng.uniform(0, 40, n)
        if radar_dims > 7:
            p[:, 7] = rng.choice([0.0, 0.1, 0.2], n)
        points.append(torch.from_numpy(p).to(device))
        k = 30
        cls = rng.integers(0, 4, k)
        sz = np.asarray(ANCHOR_SIZES)[cls] * rng.uniform(0.8, 1.2, (k, 3))
        box = np.concatenate([rng.uniform(-58, 58, (k, 1)), rng.uniform(-38, 38, (k, 1)),
                              np.asarray(ANCHOR_Z)[cls][:, None] - sz[:, 2:3] / 2 + rng.normal(0, 0.2, (k, 1)), sz,
                              rng.uniform(-math.pi, math.pi, (k, 1)), rng.normal(0, 3, (k, 2))], 1).astype(np.float32)
        gt_boxes.append(torch.from_numpy(box).to(device))
        gt_labels.append(torch.from_numpy(cls).long().to(device))
    metas = [dict(lidar2img=synthetic_lidar2img(res)) for _ in range(batch)]
    return dict(points=points, img=img.to(device), img_depth=img_depth.to(device), img_metas=metas,
                gt_bboxes_3d=gt_boxes, gt_labels_3d=gt_labels)


def _tiny_batch(batch, radar_dims, device, seed):
    t = TINY
    H, W, r = t["H"], t["W"], t["pc_range"]
    g = torch.Generator(device="cpu").manual_seed(seed)
    rng = np.random.default_rng(seed)
    img = torch.randn(batch, 6, 3, H, W, generator=g)
    img_depth = torch.zeros(batch, 6, H, W)
    mask = torch.rand(batch, 6, H, W, generator=g) < 0.1
    img_depth[mask] = torch.rand(int(mask.sum()), generator=g) * 8 + 1
    points, gt_boxes, gt_labels = [], [], []
    for _ in range(batch):
        n = int(rng.integers(*t["n_points"]))
        p = rng.normal(0, 2, (n, radar_dims)).astype(np.float32)
        p[:, 0] = rng.uniform(r[0] - 0.5, r[3] + 0.5, n); p[:, 1] = rng.uniform(r[1] - 0.5, r[4] + 0.5, n)
        p[:, 2] = rng.uniform(r[2], r[5], n)
        points.append(torch.from_numpy(p).to(device))
        k = t["n_boxes"]
        cls = rng.integers(0, 2, k)
        sz = np.asarray(t["anchor_sizes"])[cls] * rng.uniform(0.9, 1.1, (k, 3))
        box = np.concatenate([rng.uniform(r[0] + 1, r[3] - 1, (k, 1)), rng.uniform(r[1] + 1, r[4] - 1, (k, 1)),
                              np.asarray(t["anchor_z"])[cls][:, None] - sz[:, 2:3] / 2, sz,
                              rng.uniform(-math.pi, math.pi, (k, 1)), rng.normal(0, 1, (k, 2))], 1).astype(np.float32)
        gt_boxes.append(torch.from_numpy(box).to(device))
        gt_labels.append(torch.from_numpy(cls).long().to(device))
    metas = [dict(lidar2img=synthetic_lidar2img("tiny")) for _ in range(batch)]
    return dict(points=points, img=img.to(device), img_depth=img_depth.to(device), img_metas=metas,
                gt_bboxes_3d=gt_boxes, gt_labels_3d=gt_labels)


class FusionTrainStep:
    """forward_train -> sum of losses -> backward -> grad-clip 35 -> AdamW (the reference recipe,
    bevfusion.py:257-261), optionally under DistributedDataParallel (RCCL) and bf16 autocast for the
    dense convolutions (pooling, voxelisation and the losses stay fp32)."""

    def __init__(self, res="r1", batch=1, radar_dims=7, device="cuda:0", seed=0, dtype="bf16", ddp=False,
                 channels_last=True, sets=2, task="det", miopen_find=False, frames=4):
        from .mm.config import build_detector
        self.device = torch.device(device)
        if self.device.type == "cuda" and miopen_find:
            # MIOpen "find" mode: every convolution geometry is timed once over the applicable solvers instead of
            # taking the immediate-mode heuristic (41.0 -> 37.9 ms per step at R1; costs ~1 min of warm-up, so it is
            # opt-in: bench.py and the profiling scripts ask for it, the tests do not)
            torch.backends.cudnn.benchmark = True
        if self.device.type == "cuda":
            from . import ops as _ops
            if _ops.deterministic():
                # OMNIHD_DETERMINISTIC=1: every convolution pass with a kernel in this library runs on it (fixed-order sums).  What
                # stays on the library — the 7x7 stem's forward, forward / data gradient of the 59-channel depth logits and of the
                # deformable convolution's 18 offsets — measured run-to-run identical; their WEIGHT gradients were not (atomics)
                # and run on this library with the output gradient padded to a multiple of 8 (ops.wgrad_split_padded)
                # (torch.backends.cudnn.deterministic is NOT set: with that attribute the library ends up on its naive reference
                # kernels for the leftovers — 28 ms each, 273 ms per step measured; OMNIHD_DET_CUDNN=1 sets it anyway)
                if os.environ.get("OMNIHD_DET_CUDNN", "0") == "1":
                    torch.backends.cudnn.deterministic = True
        torch.manual_seed(0)                         # identical initial weights on every rank
        cfg = tiny_model_cfg(radar_dims) if res == "tiny" else model_cfg_for(res, radar_dims)
        if task == "occ":
            cfg = occ_model_cfg(cfg)
        elif task == "camera":
            cfg = camera_model_cfg(cfg)
        elif task == "triple":
            cfg = triple_model_cfg(cfg, queue_length=frames)
        model = build_detector(cfg).to(self.device)
        if channels_last:
            model = model.to(memory_format=torch.channels_last)
            for mod in model.modules():                            # point canvases written NHWC by the scatter kernel
                if hasattr(getattr(mod, "pts_middle_encoder", None), "channels_last"):
                    mod.pts_middle_encoder.channels_last = True
        model.train()
        self.raw_model = model
        self.model = model
        self.small_params = []
        if ddp:
            # The ~200 parameters of at most 4096 elements (BatchNorm weights / biases, convolution biases) stay outside the
            # reducer: inside it every one of them costs a copy launch into its bucket per step (hipMemcpyAsync: 211 launches and
            # 1.2 ms of device time per R1 step, profiles/round5/step_fp32_ddp1_before.txt).  Their gradients (0.4 MB) are
            # flattened, all-reduced in ONE message and scattered back right behind backward (``_reduce_small_params``).
            if os.environ.get("OMNIHD_DDP_SMALL_FLAT", "1") != "0":
                named = [(n, p) for n, p in model.named_parameters() if p.requires_grad and p.numel() <= 4096]
                self.small_params = [p for _, p in named]
                nn.parallel.DistributedDataParallel._set_params_and_buffers_to_ignore_for_model(model, [n for n, _ in named])
            self.model = nn.parallel.DistributedDataParallel(
                model, device_ids=[self.device.index] if self.device.type == "cuda" else None,
                broadcast_buffers=False, bucket_cap_mb=int(os.environ.get("OMNIHD_DDP_BUCKET_MB", "25")), gradient_as_bucket_view=True,
                static_graph=os.environ.get("OMNIHD_DDP_STATIC", "0") == "1")
            _broadcast_small(self.small_params, self.model.process_group)
            if os.environ.get("OMNIHD_DDP_HOOK", "1") != "0":
                from . import ops
                # bucket all-reduces on a stream that also waits for the weight-gradient side stream; side-stream weight
                # gradients written straight into the reducer's bucket views: the N > 1 step is the N = 1 step (ops/streams.py)
                ops.ddp_wgrad_overlap(self.model)
        params = [p for p in model.parameters() if p.requires_grad]
        self.opt = torch.optim.AdamW(params, lr=2e-4, weight_decay=0.05, fused=self.device.type == "cuda")
        self.params = params
        self.autocast = dtype == "bf16"
        if task == "triple":
            self.batches = [synthetic_queue(res, batch, radar_dims, self.device, seed + 1000 * i, frames=frames)
                            for i in range(sets)]
        else:
            self.batches = [synthetic_batch(res, batch, radar_dims, self.device, seed + 1000 * i) for i in range(sets)]
        if task == "occ":
            nx, ny = (int(round((cfg["pc_range"][3 + a] - cfg["pc_range"][a]) / cfg["grid"])) for a in (0, 1))
            for i, b in enumerate(self.batches):
                b["gt_occ"] = synthetic_occupancy(batch, nx, ny, 16, 12, self.device, seed + 1000 * i)
        self.i = 0
        self.last_losses = None
        self.ddp = bool(ddp)
        # A new camera calibration every step, as on the reference's own frames: its lidar2img is composed per sample through
        # the ego poses of the camera and of the LiDAR sweep (datasets/newscenes_dataset.py:203-216,
        # newscenes_devkit/newscenes_converter_final.py:346-383).  Off: the static rig of SURVEY 8(d).
        self.jitter_calibration = False
        self._jitter_rng = np.random.default_rng(seed + 77)

    def _reduce_small_params(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            ddp = self.model if isinstance(self.model, nn.parallel.DistributedDataParallel) else None
            if ddp is not None and not ddp.require_backward_grad_sync:
                return 0                                  # DDP.no_sync(): gradients accumulate locally
            return _reduce_small(self.small_params, None if ddp is None else ddp.process_group)
        return 0

    def sync_choices(self):
        """After the set-up steps of a multi-rank run: every rank takes rank 0's measured per-geometry kernel choices, so all
        ranks (and the bf16 summation order of their convolutions) agree from here on."""
        if self.ddp and self.device.type == "cuda":
            from . import ops
            return ops.sync_tuned_choices()
        return 0

    def step(self):
        from ._env import epoch_begin, epoch_end
        epoch_begin()                 # environment switches are looked up once per step from here on (omnihd_amd/_env.py)
        try:
            return self._step()
        finally:
            epoch_end()

    def _jittered(self, lidar2img):
        """The rig seen from an ego pose that moved by up to 1 degree of yaw and 0.5 m (0.05 m vertically) since the sweep."""
        a = math.radians(self._jitter_rng.uniform(-1.0, 1.0))
        T = np.eye(4)
        T[:2, :2] = [[math.cos(a), -math.sin(a)], [math.sin(a), math.cos(a)]]
        T[:3, 3] = self._jitter_rng.uniform(-0.5, 0.5, 3) * [1.0, 1.0, 0.1]
        return [np.asarray(m, dtype=np.float64) @ T for m in lidar2img]

    def _step(self):
        b = self.batches[self.i % len(self.batches)]
        self.i += 1
        if self.jitter_calibration:
            b = dict(b)
            if isinstance(b.get("img_metas"), list) and b["img_metas"] and isinstance(b["img_metas"][0], dict):
                b["img_metas"] = [dict(m, lidar2img=self._jittered(m["lidar2img"])) for m in b["img_metas"]]
        dbg = os.environ.get("OMNIHD_DEBUG_SYNC", "")      # lab switch (fault hunt, profiles/round6/fault_root_cause.txt): device syncs at phase borders
        self.opt.zero_grad(set_to_none=True)
        with torch.autocast(self.device.type, dtype=torch.bfloat16, enabled=self.autocast):
            losses = self.model(return_loss=True, **b)
        total = sum(v if torch.is_tensor(v) else sum(v) for v in losses.values())
        if "post_fwd" in dbg:
            torch.cuda.synchronize()
        total.backward()          # (ends with ops.wgrad_overlap_join: the weight gradients computed on the side stream are joined)
        if "post_bwd" in dbg:
            torch.cuda.synchronize()
        if self.small_params:
            self._reduce_small_params()
        torch.nn.utils.clip_grad_norm_(self.params, max_norm=35, norm_type=2)
        self.opt.step()
        if "post_opt" in dbg:
            torch.cuda.synchronize()
        if self.device.type == "cuda":
            from . import ops
            ops.refresh_bf16_shadows()            # one fused fp32 -> bf16 copy of all convolution weights
            ops.refresh_split_shadows()           # fp32 step: forward + data-gradient planes of every split convolution, one launch
            ops.refresh_f16_shadows()             # OMNIHD_FP32_CONV=f16: the half images of the layers on the TF32-grade form (none otherwise)
        self.last_losses = losses
        return total


_SMALL_FLAGS = {}


def _reduce_small(params, group=None):
    """Mean over the ranks of the gradients of ``params`` through ONE flat all-reduce (cat -> all_reduce -> multi-tensor copy
    back): what the reducer would do with one copy launch per parameter.

    The flat buffer has a FIXED layout — every parameter of the list in list order, zeros where this rank holds no gradient,
    followed by one "has a gradient" flag per parameter — so ranks that disagree on which parameters received a gradient (a
    data-dependent branch) still exchange buffers of one size and meaning (sizing it from ``grad is not None``
    hangs or corrupts).  A parameter without a gradient here that got one on another rank receives the mean like everybody else
    (what DistributedDataParallel does for its own parameters); one without a gradient on every rank keeps ``None``."""
    import torch.distributed as dist
    if not params:
        return 0
    dev, dt = params[0].device, params[0].dtype
    sizes = [p.numel() for p in params]
    have = [p.grad is not None for p in params]
    if not any(have) and dist.get_world_size(group) == 1:
        return 0
    zeros = {}
    parts = []
    for p, h, n in zip(params, have, sizes):
        if h:
            parts.append(p.grad.reshape(-1))
        else:
            if n not in zeros:
                zeros[n] = torch.zeros(n, dtype=dt, device=dev)
            parts.append(zeros[n])
    if all(have):
        # the common case must not cost a host-to-device copy per step (a pageable-memory upload makes the host wait for the stream)
        key = (str(dev), dt, len(params))
        ones = _SMALL_FLAGS.get(key)
        if ones is None:
            ones = _SMALL_FLAGS[key] = torch.ones(len(params), dtype=dt, device=dev)
        parts.append(ones)
    else:
        parts.append(torch.tensor([1.0 if h else 0.0 for h in have], dtype=dt, device=dev))
    flat = torch.cat(parts)
    world = dist.get_world_size(group)
    n_grad = sum(sizes)
    if world > 1:
        flat[:n_grad].mul_(1.0 / world)
    dist.all_reduce(flat, group=group)
    chunks = flat[:n_grad].split(sizes)
    if all(have):
        torch._foreach_copy_([p.grad for p in params], [c.view_as(p.grad) for c, p in zip(chunks, params)])
    else:
        anywhere = (flat[n_grad:] > 0).tolist()              # (a host read: only on the rare path where a gradient is missing)
        dst, src = [], []
        for p, h, c, a in zip(params, have, chunks, anywhere):
            if h:
                dst.append(p.grad); src.append(c.view_as(p.grad))
            elif a:
                p.grad = c.clone().view_as(p)
        if dst:
            torch._foreach_copy_(dst, src)
    return flat.numel()


def _broadcast_small(params, group=None, src=0):
    """The parameters kept outside the reducer are not covered by its construction-time broadcast: one flat broadcast instead."""
    import torch.distributed as dist
    if not params or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        flat = torch.cat([p.detach().reshape(-1) for p in params])
        dist.broadcast(flat, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
        torch._foreach_copy_([p.data for p in params], [c.view_as(p) for c, p in zip(flat.split([p.numel() for p in params]), params)])


def count_step_flops(step):
    """Dense-layer FLOPs of ONE training step of ``step`` (a FusionTrainStep), read off the module graph with forward hooks on
    one real forward: every Conv2d / ConvTranspose2d / Linear / deformable convolution contributes 2*MACs forward, the same again
    for its data gradient where its input carries a gradient, and again for its weight gradient where the weight is trainable.
    Pooling, voxelisation, BatchNorm, activations and the losses are bandwidth work and are not counted.
    Returns {"forward", "backward", "total"} in FLOPs per step (all frames of the batch)."""
    from .mm.dcn import DeformConv2dPack
    fwd = [0.0]
    bwd = [0.0]

    def macs_of(mod, x, y):
        if isinstance(mod, nn.Linear):
            return y.numel() * mod.in_features
        if isinstance(mod, DeformConv2dPack):
            return y.numel() * (mod.in_channels // mod.groups) * mod.k * mod.k
        kh, kw = mod.kernel_size
        if isinstance(mod, nn.ConvTranspose2d):
            return x.numel() * (mod.out_channels // mod.groups) * kh * kw
        return y.numel() * (mod.in_channels // mod.groups) * kh * kw

    def hook(mod, inp, out):
        x = inp[0]
        if not torch.is_tensor(x) or not torch.is_tensor(out):
            return
        f = 2.0 * macs_of(mod, x, out)
        fwd[0] += f
        w = getattr(mod, "weight", None)
        if torch.is_grad_enabled():
            bwd[0] += f * (bool(x.requires_grad) + bool(w is not None and w.requires_grad))

    handles = [m.register_forward_hook(hook) for m in step.raw_model.modules()
               if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, nn.Linear, DeformConv2dPack))]
    # The counting forward runs with every module in eval(): layer shapes and requires_grad flags are those of the training
    # pass, but no BatchNorm takes batch statistics — so it issues NO collective (naiveSyncBN / the fused pillar net all-reduce
    # only in training mode: a rank that counts alone, as bench.py's rank 0 once did, would otherwise leave dozens of
    # unmatched all-reduces in the process group's queue) and leaves the running statistics of this rank untouched.
    modes = [(m, m.training) for m in step.raw_model.modules()]
    try:
        for m, _ in modes:
            m.training = False
        b = step.batches[0]
        with torch.enable_grad(), torch.autocast(step.device.type, dtype=torch.bfloat16, enabled=step.autocast):
            losses = step.raw_model(return_loss=True, **b)
        del losses
    finally:
        for m, was in modes:
            m.training = was
        for h in handles:
            h.remove()
    return {"forward": fwd[0], "backward": bwd[0], "total": fwd[0] + bwd[0]}


def comm_report(step, iters=3):
    """What a multi-rank training step exchanges, measured / read where it happens (bench.py puts it into its JSON line as
    ``comm`` at N > 1; tests/test_distributed_cpu.py runs it over gloo):
      world_size / backend      as the process group reports them (backend "nccl" is RCCL on ROCm);
      allreduce_bytes_per_step  gradient bytes DDP all-reduces per step (every trainable parameter once, fp32);
      buckets                   DDP's gradient buckets (25 MB cap, reference: mmdet_train.py:76-80);
      syncbn_exchanges_per_step all-reduces of the naiveSyncBN layers (one forward + one backward each, 2*C floats);
      exposed_comm_ms           wall time of a step MINUS the same step under DDP.no_sync() (no gradient all-reduce): the part of
                                the communication that backward does not hide.  Run after the timed region (no_sync steps let the
                                ranks' weights drift apart)."""
    import time
    import torch.distributed as dist
    from .mm.sync_bn import _NaiveSyncBN
    if not (dist.is_available() and dist.is_initialized()):
        return None
    params = [p for p in step.raw_model.parameters() if p.requires_grad]
    nbytes = sum(p.numel() * p.element_size() for p in params)
    small = getattr(step, "small_params", [])
    rep = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "allreduce_bytes_per_step": int(nbytes),
           "small_params_flat_allreduce": {"tensors": len(small), "bytes": int(sum(p.numel() * p.element_size() for p in small))},
           "bucket_cap_mb": 25, "buckets": None, "exposed_comm_ms": None,
           "syncbn_exchanges_per_step": 2 * sum(1 for m in step.raw_model.modules()
                                                if (isinstance(m, _NaiveSyncBN) or isinstance(m, nn.SyncBatchNorm)) and m.training)}
    ddp = step.model if isinstance(step.model, nn.parallel.DistributedDataParallel) else None
    if ddp is None:
        return rep

    def sync():
        if step.device.type == "cuda":
            torch.cuda.synchronize(step.device)

    def timed(no_sync):
        ts = []
        for _ in range(iters):
            dist.barrier()
            sync()
            t0 = time.perf_counter()
            if no_sync:
                with ddp.no_sync():
                    step.step()
            else:
                step.step()
            sync()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]

    t_sync = timed(False)
    try:
        data = ddp._get_ddp_logging_data()
        # after its first backward DDP re-buckets in the order the gradients arrived: those are the buckets that are reduced
        sizes = str(data.get("rebuilt_bucket_sizes", "") if data.get("has_rebuilt_buckets") else "").strip() \
            or str(data.get("bucket_sizes", "")).strip()
        if sizes:
            rep["buckets"] = len([s for s in sizes.split(",") if s.strip()])
    except Exception:
        pass
    if rep["buckets"] is None:
        rep["buckets"] = max(1, -(-nbytes // (25 << 20)))
    t_nosync = timed(True)
    rep["step_ms"] = round(t_sync * 1e3, 3)
    rep["step_no_allreduce_ms"] = round(t_nosync * 1e3, 3)
    rep["exposed_comm_ms"] = round(max(t_sync - t_nosync, 0.0) * 1e3, 3)
    return rep


def syncbn_exchange_probe(step, iters=20):
    """The naiveSyncBN statistic exchanges of ONE training step of ``step`` (reference ops/norm.py:55-82: per layer one
    exchange of the 2*C statistics forward and one of the 2*C gradient sums backward; here one all-reduce each), issued back to
    back on the current stream in the default process group: wall time per step's worth of exchanges and per call.  Inside a
    one-rank group the step itself skips them (plain BatchNorm, norm.py:58), so this is how one GPU measures their launch cost."""
    import time
    import torch.distributed as dist
    from .mm.sync_bn import _NaiveSyncBN
    if not (dist.is_available() and dist.is_initialized()):
        return None
    chans = [m.num_features for m in step.raw_model.modules()
             if (isinstance(m, _NaiveSyncBN) or isinstance(m, nn.SyncBatchNorm)) and m.training]
    if not chans:
        return {"exchanges_per_step": 0, "total_us_per_step": 0.0, "per_call_us": 0.0}
    bufs = [torch.zeros(2 * c, dtype=torch.float32, device=step.device) for c in chans] * 2        # forward + backward
    def sync():
        if step.device.type == "cuda":
            torch.cuda.synchronize(step.device)
    for b in bufs:
        dist.all_reduce(b)
    sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        for b in bufs:
            dist.all_reduce(b)
    sync()
    dt = (time.perf_counter() - t0) / iters
    return {"exchanges_per_step": len(bufs), "total_us_per_step": round(dt * 1e6, 1), "per_call_us": round(dt * 1e6 / len(bufs), 2),
            "message_floats": [2 * c for c in sorted(set(chans))],
            "note": "all-reduces issued back to back by one thread, as in the step (the radar branch runs in line since round 6)"}

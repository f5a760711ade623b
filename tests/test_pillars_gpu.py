"""GPU parity of the point-stream-only detectors (``MVXFasterRCNN`` with radar PFN, RCFusion radar
PFN or LiDAR ``HardVFE``): the HIP voxeliser at the LiDAR stream's sizes (64 points per pillar,
~120k points) bit-exact against the sequential oracle, and one tiny step through the HIP operators
against the same weights on the CPU over the oracle operators (1e-3 relative, fp32)."""
import contextlib
import os

import numpy as np
import pytest
import torch

from tests.test_radar_gpu import RNG6, VS, check, radar_cloud

pytestmark = pytest.mark.gpu


def _close(a, b, tol=1e-3):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-6) <= tol


def test_voxelize_lidar_sized_cloud_64_points_per_pillar(cuda):
    rng = np.random.default_rng(64)
    pts = radar_cloud(rng, 120000, f=4, spread=1.02)
    pts[:20000, :2] *= 0.03                              # a dense core: its pillars overflow 64 points
    w = check(cuda, pts, VS, RNG6, 64, 30000)
    assert w[2].max() == 64 and len(w[1]) == 30000       # both caps are hit
    w = check(cuda, pts[:30000], VS, RNG6, 64, 40000)
    assert 0 < len(w[1]) < 40000


def _run(device, use_oracle, stream):
    from omnihd_amd import harness
    from omnihd_amd.mm.config import build_detector
    from oracle.torch_shim import oracle_ops
    with (oracle_ops() if use_oracle else contextlib.nullcontext()):
        torch.manual_seed(1)
        m = build_detector(harness.pillars_model_cfg(harness.tiny_model_cfg(7), stream)).to(device).eval()
        b = harness.synthetic_batch("tiny", 2, 7, device, 0)
        pts = [p[:, :4].contiguous() for p in b["points"]] if stream == "lidar" else b["points"]
        feats = m.extract_feat(pts, None, b["img_metas"])[1]
        losses = m(return_loss=True, points=pts, img_metas=b["img_metas"], gt_bboxes_3d=b["gt_bboxes_3d"],
                   gt_labels_3d=b["gt_labels_3d"])
        total = sum(v[0] if isinstance(v, list) else v for v in losses.values())
        total.backward()
        torch.nn.init.constant_(m.pts_bbox_head.conv_cls.bias, 0.0)
        dets = m(return_loss=False, points=[pts], img_metas=[b["img_metas"]])
        return dict(feat=feats[0].detach().cpu(),
                    losses={k: float((v[0] if isinstance(v, list) else v).detach()) for k, v in losses.items()},
                    grads={n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None},
                    n_det=[len(r["pts_bbox"]["boxes_3d"]) for r in dets],
                    scores=[r["pts_bbox"]["scores_3d"] for r in dets])


@pytest.mark.parametrize("stream", ["radar", "rcfusion", "lidar"])
def test_stream_only_tiny_step_hip_ops_match_oracle_ops(cuda, stream, monkeypatch):
    monkeypatch.setenv("OMNIHD_FP32_CONV", "miopen")     # operator parity: library fp32 convolutions (see test_detector_gpu._grads_agree)
    got, want = _run(cuda, False, stream), _run("cpu", True, stream)
    assert got["feat"].shape == want["feat"].shape and _close(got["feat"], want["feat"])
    for k, v in want["losses"].items():
        assert abs(got["losses"][k] - v) <= 1e-3 * max(abs(v), 1e-3), (k, got["losses"][k], v)
    assert set(got["grads"]) == set(want["grads"])
    enc = [n for n in want["grads"] if n.startswith("pts_voxel_encoder") and n.endswith("linear.weight")
           or n.endswith("linear1.weight")]
    assert enc
    for n in enc + ["pts_backbone.blocks.0.0.weight", "pts_bbox_head.conv_reg.weight"]:
        assert _close(got["grads"][n], want["grads"][n], 2e-3), n
    # test-time path: same number of detections with the same score profile (box-by-box identity is not asserted:
    # near-tied scores may order differently when the logits differ in the last bits)
    assert got["n_det"] == want["n_det"]
    for sa, sb in zip(got["scores"], want["scores"]):
        assert _close(sa.sort()[0], sb.sort()[0])


def test_packed_hard_vfe_on_the_gpu_matches_the_dense_form_at_lidar_sizes(cuda):
    import copy
    from omnihd_amd import ops
    from omnihd_amd.mm.hard_vfe import HardVFE
    rng = np.random.default_rng(5)
    pts = radar_cloud(rng, 120000, f=4, spread=1.0)
    vox, coors, num = ops.hard_voxelize(torch.from_numpy(pts).to(cuda), VS, RNG6, 64, 30000)
    coors = torch.nn.functional.pad(coors, (1, 0))
    torch.manual_seed(0)
    dense = HardVFE(in_channels=4, feat_channels=[64, 64], with_cluster_center=True, with_voxel_center=True, voxel_size=VS,
                    point_cloud_range=RNG6, norm_cfg=dict(type="naiveSyncBN1d", eps=1e-3, momentum=0.01), packed=False).to(cuda)
    packed = copy.deepcopy(dense)
    packed.packed = True
    for mode in ("eval", "train"):
        getattr(dense, mode)(); getattr(packed, mode)()
        a, b = dense(vox, num, coors), packed(vox, num, coors, max_real_points=pts.shape[0])
        assert _close(b, a, 1e-4), mode
        a.sum().backward(); b.sum().backward()
    for (n_, p), q in zip(dense.named_parameters(), packed.parameters()):
        assert _close(q.grad, p.grad, 5e-2), n_          # the dense float32 gradient is the loose side (see the CPU test)

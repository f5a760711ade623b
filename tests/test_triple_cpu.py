"""CPU tests of the triple-modal temporal composition (BASELINE.json configs[4]; SURVEY.md D11: no
reference counterpart, parity unpinned by construction).  Checked here: the BEV resampling between
ego frames by known answers, the module graph / state-dict names against the reference ingredients,
the queue semantics (history frames in eval mode without gradients, one flat batch) and one tiny
training + test step with the HIP operators routed to the CPU oracle (tests only)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F


def _warp(bev, delta, pc_range):
    from projects.mmdet3d_plugin.bevfusion.detectors.bevf_triple_temporal import bev_warp_theta
    theta = torch.from_numpy(bev_warp_theta(delta, pc_range)).float()[None]
    grid = F.affine_grid(theta, list(bev.shape), align_corners=False)
    return F.grid_sample(bev, grid, mode="bilinear", padding_mode="zeros", align_corners=False)


def test_bev_warp_known_answers():
    pc = [-8.0, -6.0, -1.0, 8.0, 6.0, 1.0]                   # 16 x 12 cells of 1 m; cell (iy, ix) centre = (-7.5+ix, -5.5+iy)
    bev = torch.zeros(1, 1, 12, 16)
    bev[0, 0, 4, 5] = 1.0                                    # a hot cell at x = -2.5, y = -1.5 in the history frame
    assert torch.allclose(_warp(bev, (0.0, 0.0, 0.0), pc), bev, atol=1e-5)
    out = _warp(bev, (3.0, -2.0, 0.0), pc)                   # ego moved: the point is now at x = 0.5, y = -3.5
    want = torch.zeros_like(bev); want[0, 0, 2, 8] = 1.0
    assert torch.allclose(out, want, atol=1e-5)
    out = _warp(bev, (0.5, 0.0, 0.0), pc)                    # half a cell: split between two cells
    assert abs(float(out[0, 0, 4, 5]) - 0.5) < 1e-5 and abs(float(out[0, 0, 4, 6]) - 0.5) < 1e-5
    out = _warp(bev, (100.0, 0.0, 0.0), pc)                  # moved out of the range: zeros, not clamped copies
    assert float(out.abs().sum()) == 0.0
    sq = [-6.0, -6.0, -1.0, 6.0, 6.0, 1.0]                   # square range: a quarter turn maps cells onto cells
    b2 = torch.zeros(1, 1, 12, 12); b2[0, 0, 6, 9] = 1.0     # x = 3.5, y = 0.5
    out = _warp(b2, (0.0, 0.0, math.pi / 2), sq)             # R(90 deg) (3.5, 0.5) = (-0.5, 3.5)
    want = torch.zeros_like(b2); want[0, 0, 9, 5] = 1.0
    assert torch.allclose(out, want, atol=1e-5)


def test_full_size_config_is_the_reference_ingredients():
    from omnihd_amd import harness
    from omnihd_amd.mm.config import build_detector
    base = harness.reference_model_cfg()
    m = build_detector(harness.triple_model_cfg(base, queue_length=4))
    fusion = build_detector(base)
    lidar = build_detector(harness.pillars_model_cfg(base, "lidar"))
    sd, fsd, lsd = m.state_dict(), fusion.state_dict(), lidar.state_dict()
    # every tensor of the fusion detector is there under the same name (reduc_conv.conv.weight widened by lic)
    for k, v in fsd.items():
        assert k in sd and (sd[k].shape == v.shape or k == "reduc_conv.conv.weight"), k
    assert sd["reduc_conv.conv.weight"].shape == (384, 256 + 2 * 384, 3, 3)
    # every stream tensor of the LiDAR PointPillars detector is there under the lidar_stream. prefix
    for k, v in lsd.items():
        if not k.startswith("pts_bbox_head"):
            assert sd["lidar_stream." + k].shape == v.shape, k
    assert not any(k.startswith("lidar_stream.pts_bbox_head") for k in sd)
    assert sd["temporal_conv.conv.weight"].shape == (384, 4 * 384, 3, 3)
    extra = set(sd) - set(fsd) - {"lidar_stream." + k for k in lsd}
    assert all(k.startswith("temporal_conv.") for k in extra), sorted(extra)[:5]
    assert m.lidar_stream.pts_voxel_layer.max_num_points == 64 and m.queue_length == 4
    with pytest.raises(ValueError):
        build_detector(dict(harness.triple_model_cfg(base), lidar_stream=None))


def test_tiny_queue_step_over_the_oracle_and_queue_semantics():
    from omnihd_amd.harness import FusionTrainStep
    from oracle.torch_shim import oracle_ops
    torch.set_num_threads(4)
    with oracle_ops():
        st = FusionTrainStep(res="tiny", batch=2, radar_dims=7, device="cpu", dtype="fp32", channels_last=False, sets=1,
                             task="triple", frames=3)
        m, b = st.raw_model, st.batches[0]
        assert b["img"].shape[:3] == (2, 3, 6) and len(b["points"]) == 2 and len(b["lidar_points"][0]) == 3
        # history frames: eval mode (BatchNorm statistics untouched), no graph, training mode restored
        bn = m.reduc_conv.bn
        before = bn.running_mean.clone()
        hist = m._history_bev(b["points"], b["lidar_points"], b["img"], b["img_metas"])
        assert hist.shape == (2, 2 * 384, 12, 16) and not hist.requires_grad and m.training
        assert torch.equal(bn.running_mean, before)
        # one flat batch == frame by frame (eval mode: samples are independent)
        m.eval()
        with torch.no_grad():
            one = m.extract_feat([b["points"][1][0]], b["img"][1:2, 0], [dict(b["img_metas"][1][0])],
                                 lidar_points=[b["lidar_points"][1][0]])["pts_feats"][0]
        want = _warp(one, b["img_metas"][1][0]["ego_delta"], m._pc_range)
        assert torch.allclose(hist[1, :384], want[0], atol=1e-4)
        m.train()
        l0 = float(st.step().detach())
        missing = [n for n, p in m.named_parameters() if p.requires_grad and p.grad is None]
        assert not missing, missing
        assert float(m.lidar_stream.pts_voxel_encoder.vfe_layers[0].linear.weight.grad.abs().sum()) > 0
        assert float(m.temporal_conv.conv.weight.grad[:, :2 * 384].abs().sum()) > 0      # history channels are used
        for _ in range(3):
            l1 = float(st.step().detach())
        assert set(st.last_losses) == {"loss_cls", "loss_bbox", "loss_dir", "img_depth_loss"}
        assert math.isfinite(l0) and math.isfinite(l1) and l1 < l0
        # the test-time entry point
        m.eval()
        torch.nn.init.constant_(m.pts_bbox_head.conv_cls.bias, 0.0)
        out = m(return_loss=False, points=[b["points"]], img_metas=[b["img_metas"]], img=[b["img"]],
                lidar_points=[b["lidar_points"]])
        assert len(out) == 2 and all(len(r["pts_bbox"]["boxes_3d"]) > 0 for r in out)
        # wrong queue length / ragged queue are refused
        with pytest.raises(ValueError, match="queue of 2"):
            m.extract_queue_feat([p[:2] for p in b["points"]], [p[:2] for p in b["lidar_points"]], b["img"][:, :2],
                                 [q[:2] for q in b["img_metas"]])
        with pytest.raises(ValueError, match="lidar_points"):
            m.extract_queue_feat(b["points"], [p[:2] for p in b["lidar_points"]], b["img"], b["img_metas"])
    assert np.isfinite(l1)

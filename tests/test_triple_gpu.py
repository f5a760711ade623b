"""GPU checks of the triple-modal temporal composition (BASELINE.json configs[4]; no reference
counterpart): the tiny model through the HIP operators against the same weights on the
CPU over the oracle operators (1e-3 relative, fp32), and the full-size configuration at bs=2 with a
4-frame queue as one bf16 training step."""
import contextlib
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(a, b, tol=1e-3):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-6) <= tol


def _run(device, use_oracle):
    from omnihd_amd.harness import FusionTrainStep
    from oracle.torch_shim import oracle_ops
    with (oracle_ops() if use_oracle else contextlib.nullcontext()):
        st = FusionTrainStep(res="tiny", batch=2, radar_dims=7, device=device, seed=3, dtype="fp32", channels_last=False,
                             sets=1, task="triple", frames=3)
        m, b = st.raw_model, st.batches[0]
        m.eval()                      # BN in eval everywhere: the comparison is about the operators
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        hist = m._history_bev(b["points"], b["lidar_points"], b["img"], b["img_metas"])
        losses = m(return_loss=True, **b)
        total = sum(v[0] if isinstance(v, list) else v for v in losses.values())
        total.backward()
        return dict(hist=hist.cpu(), losses={k: float((v[0] if isinstance(v, list) else v).detach()) for k, v in losses.items()},
                    grads={n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None})


def test_tiny_queue_hip_ops_match_oracle_ops(cuda, monkeypatch):
    # an OPERATOR parity test: the dense fp32 convolutions are pinned to the library kernels here (with the fp32-grade split kernels
    # single ReLU masks of these tiny maps flip and move gradient entries by percent — tests/test_detector_gpu.py::_grads_agree)
    monkeypatch.setenv("OMNIHD_FP32_CONV", "miopen")
    got, want = _run(cuda, False), _run("cpu", True)
    assert got["hist"].shape == want["hist"].shape and _close(got["hist"], want["hist"])
    for k, v in want["losses"].items():
        assert abs(got["losses"][k] - v) <= 1e-3 * max(abs(v), 1e-3), (k, got["losses"][k], v)
    assert set(got["grads"]) == set(want["grads"])
    for n in ("temporal_conv.conv.weight", "reduc_conv.conv.weight", "lidar_stream.pts_voxel_encoder.vfe_layers.0.linear.weight",
              "lidar_stream.pts_backbone.blocks.0.0.weight", "pts_voxel_encoder.pfn_layers.0.linear.weight",
              "lift_splat_shot_vis.bevencode.0.weight"):
        # (5e-3 as in tests/test_detector_gpu.py: the dense convolutions of the fp32 path run on the fp32-grade split kernels or on
        # MIOpen by a per-geometry measurement, 1e-5 apart; behind ReLUs that moves single gradient entries)
        assert _close(got["grads"][n], want["grads"][n], 5e-3), n


def test_full_size_bs2_four_frame_bf16_step(cuda):
    """BASELINE.json configs[4] at full size: camera + radar + LiDAR, 4-frame queue, bs = 2 per GPU, bf16.  In the verified set
    since round 3: with MIOpen's records for its batch sizes in omnihd-scenes_amd/miopen_db (seeded by the ``cuda`` fixture)
    the first step takes ~5 s instead of 7.7 min (profiles/round3/configs4_*)."""
    from omnihd_amd.harness import FusionTrainStep
    st = FusionTrainStep(res="r1", batch=2, radar_dims=7, device=cuda, dtype="bf16", sets=1, task="triple", frames=4)
    b = st.batches[0]
    assert b["img"].shape == (2, 4, 6, 3, 256, 704) and b["lidar_points"][0][0].shape == (120000, 4)
    first = float(st.step().detach())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        last = st.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    last = float(last.detach())
    print(f"\ntriple-modal 4-frame bs=2 step: {ms:.1f} ms ({2 / ms * 1e3:.1f} frames/s), loss {first:.3f} -> {last:.3f}, "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    assert all(torch.isfinite(p.grad).all() for p in st.params if p.grad is not None)
    assert first == first and last == last and last < 1e6
    assert set(st.last_losses) == {"loss_cls", "loss_bbox", "loss_dir", "img_depth_loss"}

"""GPU parity of the assembled detector: the scaled-down BEVFUSION_depth (same module graph as the
reference config) run with the HIP operators on the GPU vs the same weights and inputs on the CPU
with the operators routed to the oracle.  Tolerance: 1e-3 relative (north_star) on BEV features,
head outputs and losses; fp32 everywhere (no autocast) so that only summation order differs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(device, use_oracle, variant="bevfusion"):
    from omnihd_amd.harness import FusionTrainStep
    from oracle.torch_shim import oracle_ops
    import contextlib
    ctx = oracle_ops() if use_oracle else contextlib.nullcontext()
    with ctx:
        st = FusionTrainStep(res="tiny", batch=2, radar_dims=7, device=device, seed=3, dtype="fp32", channels_last=False, sets=1)
        m, b = st.raw_model, st.batches[0]
        if variant == "rcfusion":
            # RCFusion_FasterRCNN = the same streams with RadarPillarFeatureNet and Cross_Modal_Fusion
            # (rcfusion/detectors/rcfusion_faster_rcnn.py:141-144, BEVCross_modal_attention.py:6-43)
            from omnihd_amd import harness
            from omnihd_amd.mm.config import build_detector
            c = harness.tiny_model_cfg(7)
            c["type"] = "RCFusion_FasterRCNN"
            c.pop("lc_fusion")
            c["rc_fusion"] = "cross_attention"
            c["pts_voxel_encoder"].update(type="RadarPillarFeatureNet", with_velocity_snr_center=True)
            torch.manual_seed(0)
            m = build_detector(c).to(device)
        m.eval()                      # BN in eval: the comparison is about the operators, not batch statistics
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        fd = m.extract_feat(b["points"], img=b["img"], img_metas=b["img_metas"])
        outs = m.pts_bbox_head(fd["pts_feats"])
        losses = m.pts_bbox_head.loss(*outs, b["gt_bboxes_3d"], b["gt_labels_3d"], b["img_metas"])
        depth_loss, _ = m.lift_splat_shot_vis.get_depth_loss(b["img_depth"], fd["depth_dist"], "kld")
        total = sum(v[0] for v in losses.values()) + depth_loss
        total.backward()
        grads = {n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None}
        res = dict(bev=fd["pts_feats"][0].detach().cpu(), depth=fd["depth_dist"].detach().cpu(),
                   cls=outs[0][0].detach().cpu(), reg=outs[1][0].detach().cpu(),
                   losses={k: float(v[0]) for k, v in losses.items()}, depth_loss=float(depth_loss), grads=grads)
    return res


_CPU_FORWARD = {}          # res -> outputs of the oracle-backed CPU forward (the same for every convolution policy)


def _close(a, b, tol=1e-3):
    scale = max(float(b.abs().max()), 1e-6)
    return float((a - b).abs().max()) / scale <= tol


CONV_POLICIES = ["miopen", "split"]


def _grads_agree(gpu, cpu, names, policy):
    """Gradients of the named parameters of the TINY detector, GPU (HIP operators) vs CPU (oracle operators).
    miopen: the dense convolutions on MIOpen's fp32 kernels agree with the CPU's to ~1e-6, no ReLU mask of these tiny maps flips,
    and every entry is held to 5e-3 of the largest (measured: worst 2.7e-3, 192 of 194 tensors below 1e-3).
    split: the fp32-grade kernels differ from an fp32 convolution by ~5e-6 per layer; on feature maps of a few pixels that flips
    single ReLU masks, and one flipped mask moves single ENTRIES of the gradients upstream by percents (measured over all 194
    tensors, profiles/round5/tiny_grads_split.txt: worst entry 4.2e-2, worst relative L2 7.5e-3, both in the image backbone; 142
    tensors below 1e-3 entry-wise).  What is asserted for this policy is therefore the tensor as a whole — relative L2 error
    <= 2e-2 and cosine >= 0.9995 — and the entry-wise 1e-3 bound lives where masks cannot flip the comparison:
    tests/test_stage_gradients_gpu.py (every dense stage at FULL size on identical inputs, both policies) and
    tests/test_conv_split_gpu.py (1e-4 per kernel)."""
    for n in names:
        a, b = gpu["grads"][n].double(), cpu["grads"][n].double()
        if policy == "miopen":
            assert _close(a, b, 5e-3), n
        else:
            l2 = float((a - b).norm()) / max(float(b.norm()), 1e-30)
            cos = float((a * b).sum()) / max(float(a.norm()) * float(b.norm()), 1e-30)
            assert l2 <= 2e-2 and cos >= 0.9995, (n, l2, cos)


@pytest.mark.parametrize("policy", CONV_POLICIES)
def test_tiny_detector_hip_ops_match_oracle_ops(cuda, policy, monkeypatch):
    monkeypatch.setenv("OMNIHD_FP32_CONV", policy)
    gpu = _run("cuda:0", use_oracle=False)
    cpu = _run("cpu", use_oracle=True)
    assert _close(gpu["depth"], cpu["depth"]), "depth distribution"
    assert _close(gpu["bev"], cpu["bev"]), "fused BEV feature"
    assert _close(gpu["cls"], cpu["cls"]) and _close(gpu["reg"], cpu["reg"]), "head outputs (box regressions)"
    for k in cpu["losses"]:
        assert abs(gpu["losses"][k] - cpu["losses"][k]) <= 1e-3 * max(abs(cpu["losses"][k]), 1e-3), k
    assert abs(gpu["depth_loss"] - cpu["depth_loss"]) <= 1e-3 * abs(cpu["depth_loss"])
    # gradients through the HIP backward kernels (pooling, pillar gather) reach the same values
    _grads_agree(gpu, cpu, ["lift_splat_shot_vis.camencode.depthnet.context_conv.weight", "lift_splat_shot_vis.camencode.depthnet.depth_conv.5.weight",
                            "pts_voxel_encoder.pfn_layers.0.linear.weight", "img_neck.reduc_conv.conv.weight", "reduc_conv.conv.weight"], policy)


@pytest.mark.parametrize("policy", CONV_POLICIES)
def test_tiny_detector_fp32_backward_with_allow_tf32_switched_off(cuda, policy, monkeypatch):
    """the reference's `close_tf32` switch (tools/train.py:148-153) sets torch.backends.cudnn.allow_tf32 =
    False.  Round 1 suspected MIOpen's fp32 backward kernels under that setting (1e-2 off); the error was traced to torch's
    channels-last BatchNorm backward instead, which the product no longer runs.  The same parity bounds as the
    default setting must hold with the switch off."""
    monkeypatch.setenv("OMNIHD_FP32_CONV", policy)
    torch.backends.cudnn.allow_tf32 = False
    try:
        gpu = _run("cuda:0", use_oracle=False)
    finally:
        torch.backends.cudnn.allow_tf32 = True
    cpu = _run("cpu", use_oracle=True)
    assert _close(gpu["bev"], cpu["bev"]) and _close(gpu["reg"], cpu["reg"]) and _close(gpu["depth"], cpu["depth"])
    _grads_agree(gpu, cpu, ["lift_splat_shot_vis.camencode.depthnet.context_conv.weight", "lift_splat_shot_vis.camencode.depthnet.depth_conv.5.weight",
                            "img_neck.reduc_conv.conv.weight", "reduc_conv.conv.weight", "lift_splat_shot_vis.bevencode.0.weight"], policy)


@pytest.mark.parametrize("policy", CONV_POLICIES)
def test_tiny_rcfusion_detector_hip_ops_match_oracle_ops(cuda, policy, monkeypatch):
    """RCFusion_FasterRCNN (RadarPillarFeatureNet + Cross_Modal_Fusion) on the GPU for the first time:
    HIP operators vs the same weights on the CPU over the oracle operators, fp32, 1e-3."""
    monkeypatch.setenv("OMNIHD_FP32_CONV", policy)
    gpu = _run("cuda:0", use_oracle=False, variant="rcfusion")
    cpu = _run("cpu", use_oracle=True, variant="rcfusion")
    assert _close(gpu["depth"], cpu["depth"]) and _close(gpu["bev"], cpu["bev"]), "distribution / cross-modal BEV feature"
    assert _close(gpu["cls"], cpu["cls"]) and _close(gpu["reg"], cpu["reg"]), "head outputs"
    for k in cpu["losses"]:
        assert abs(gpu["losses"][k] - cpu["losses"][k]) <= 1e-3 * max(abs(cpu["losses"][k]), 1e-3), k
    assert abs(gpu["depth_loss"] - cpu["depth_loss"]) <= 1e-3 * abs(cpu["depth_loss"])
    _grads_agree(gpu, cpu, ["cross_attention.att_img.0.weight", "cross_attention.att_radar.0.weight", "cross_attention.reduce_mixBEV.conv.weight",
                            "pts_voxel_encoder.pfn_layers.0.linear1.weight", "lift_splat_shot_vis.camencode.depthnet.context_conv.weight"], policy)


@pytest.mark.parametrize("res,policy", [("r1", "split"), ("r1", "miopen"), ("r2", "split")])
def test_full_size_fp32_forward_matches_the_oracle_ops_run(cuda, res, policy, monkeypatch):
    """(``policy``: the dense convolutions on the fp32-grade split kernels of this library / on MIOpen's fp32 kernels.)
    VERDICT round 2 #5(c): north_star's 1e-3 on the fused BEV feature and the box regressions at the BASELINE size, not only
    on the tiny model — one fp32 forward of the reference config at R1 (6 x 256 x 704, BatchNorm in inference mode, seeded
    weights) on the GPU through the HIP path vs the same weights on the CPU with the operators routed to the oracle.
    ``res`` = "r2": the same at the repo's own resolution (6 x 544 x 960 images, 8-channel radar points; bevfusion.py:28,164)."""
    import contextlib
    from omnihd_amd.harness import FusionTrainStep
    from oracle.torch_shim import oracle_ops
    monkeypatch.setenv("OMNIHD_FP32_CONV", policy)
    out = {}
    for device, use_oracle in (("cuda:0", False), ("cpu", True)):
        if device == "cpu" and res in _CPU_FORWARD:
            out[device] = _CPU_FORWARD[res]
            continue
        with (oracle_ops() if use_oracle else contextlib.nullcontext()):
            st = FusionTrainStep(res=res, batch=1, radar_dims=7 if res == "r1" else 8, device=device, seed=5, dtype="fp32",
                                 channels_last=device != "cpu", sets=1)
            m, b = st.raw_model, st.batches[0]
            m.eval()
            if device == "cpu":
                torch.set_num_threads(min(32, __import__("os").cpu_count() or 8))
            with torch.no_grad():
                fd = m.extract_feat(b["points"], img=b["img"], img_metas=b["img_metas"])
                cls, reg, dirs = m.pts_bbox_head(fd["pts_feats"])
            out[device] = dict(bev=fd["pts_feats"][0].float().cpu(), depth=fd["depth_dist"].float().cpu(),
                               cls=cls[0].float().cpu(), reg=reg[0].float().cpu())
            if device == "cpu":
                _CPU_FORWARD[res] = out[device]              # the CPU run is the same for both policies: computed once
            del st, m, fd
    gpu, cpu = out["cuda:0"], out["cpu"]
    assert gpu["bev"].shape == (1, 384, 160, 240) and gpu["reg"].shape == (1, 72, 160, 240)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    # (the depth distribution is not one of north_star's 1e-3 quantities: with BatchNorm in inference mode on random-init
    # weights DepthNet's logits reach several hundred and its softmax is nearly one-hot, so a 1e-5 relative difference in a
    # logit moves a probability by 1e-3 — scripts/lab/split_vs_miopen_r1.py: 1.35e-3 between the split kernels and MIOpen's
    # fp32 kernels in this setting, 2e-4 with batch statistics; the BEV feature built from it stays within 1e-4)
    assert _close(gpu["depth"], cpu["depth"], 5e-3), "depth distribution"
    assert _close(gpu["bev"], cpu["bev"]) and rel(gpu["bev"], cpu["bev"]) <= 1e-3, ("fused BEV feature", rel(gpu["bev"], cpu["bev"]))
    assert _close(gpu["reg"], cpu["reg"]) and _close(gpu["cls"], cpu["cls"]), "box regressions / class logits"


_TRAIN_CPU = {}
GRAD_NAMES = ["lift_splat_shot_vis.bevencode.0.weight", "lift_splat_shot_vis.bevencode.9.weight",
              "lift_splat_shot_vis.camencode.depthnet.context_conv.weight", "lift_splat_shot_vis.camencode.depthnet.depth_conv.5.weight",
              "lift_splat_shot_vis.camencode.depthnet.reduce_conv.0.weight", "img_neck.reduc_conv.conv.weight", "reduc_conv.conv.weight",
              "pts_voxel_encoder.pfn_layers.0.linear.weight", "pts_backbone.blocks.0.0.weight", "pts_bbox_head.conv_reg.weight",
              "img_backbone.layer3.0.conv1.weight"]


def _train_pass(device, use_oracle, relu_open=False):
    """One fp32 TRAINING-mode forward + backward of the reference config at R1 (BatchNorm on batch statistics, dropout off):
    losses, depth distribution, fused BEV feature and the gradients of GRAD_NAMES.  ``relu_open``: every BatchNorm gets weight 1
    and bias +6, so that (almost) no ReLU behind a BatchNorm clips — the same module graph and kernels with piecewise-linear
    units pinned to their linear side (see the test below for why)."""
    import contextlib
    from omnihd_amd.harness import FusionTrainStep
    from oracle.torch_shim import oracle_ops
    with (oracle_ops() if use_oracle else contextlib.nullcontext()):
        st = FusionTrainStep(res="r1", batch=1, radar_dims=7, device=device, seed=5, dtype="fp32", channels_last=device != "cpu", sets=1)
        m, b = st.raw_model, st.batches[0]
        m.train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if relu_open and isinstance(mod, torch.nn.modules.batchnorm._BatchNorm) and mod.affine:
                with torch.no_grad():
                    mod.weight.fill_(1.0)
                    mod.bias.fill_(6.0)
        if device == "cpu":
            torch.set_num_threads(min(32, __import__("os").cpu_count() or 8))
        fd = m.extract_feat(b["points"], img=b["img"], img_metas=b["img_metas"])
        outs = m.pts_bbox_head(fd["pts_feats"])
        losses = m.pts_bbox_head.loss(*outs, b["gt_bboxes_3d"], b["gt_labels_3d"], b["img_metas"])
        depth_loss, _ = m.lift_splat_shot_vis.get_depth_loss(b["img_depth"], fd["depth_dist"], "kld")
        (sum(v[0] for v in losses.values()) + depth_loss).backward()
        named = dict(m.named_parameters())
        res = dict(depth=fd["depth_dist"].detach().float().cpu(), bev=fd["pts_feats"][0].detach().float().cpu(),
                   reg=outs[1][0].detach().float().cpu(), losses={k: float(v[0]) for k, v in losses.items()},
                   depth_loss=float(depth_loss), grads={n: named[n].grad.detach().float().cpu() for n in GRAD_NAMES})
        del st, m, fd, outs
    return res


@pytest.mark.parametrize("policy", ["split", "miopen"])
def test_full_size_r1_training_pass_gradients_match_the_oracle_ops_run(cuda, policy, monkeypatch):
    """one fp32 TRAINING-mode forward + backward of the reference config at R1 (6 x 256 x 704, BatchNorm on
    batch statistics) through the HIP path against the same weights on the CPU with the operators routed to the oracle.  Held to
    1e-3 (relative L2 for tensors, relative for scalars): the depth distribution with train-mode BatchNorm (measured 1.2e-4), the
    fused BEV feature, the box regressions and the four losses.
    The GRADIENTS of the assembled detector are printed, not held to 1e-3, because they cannot be for ANY pair of
    implementations: two runs of the same GPU configuration differ by 1e-2 in them (profiles/round4/determinism_pass.txt: MIOpen's
    fp32 kernels for the strided convolutions accumulate with atomics, 5e-8 on the first such layer's output, amplified by the
    randomly initialised network — ReLU-mask flips count as sqrt(fraction) in a relative L2 norm — to 1e-4 on the depth
    distribution and 1–3e-2 on the image backbone's weight gradients), and MIOpen-only vs the CPU shows the same 5e-3…3e-2
    (profiles/round4/train_pass_parity.txt).  The 1e-3 gradient gate is tests/test_stage_gradients_gpu.py: every dense stage
    at full size on identical inputs, both policies."""
    monkeypatch.setenv("OMNIHD_FP32_CONV", policy)
    torch.backends.cudnn.allow_tf32 = False
    try:
        gpu = _train_pass("cuda:0", use_oracle=False)
    finally:
        torch.backends.cudnn.allow_tf32 = True
    if "r1" not in _TRAIN_CPU:
        _TRAIN_CPU["r1"] = _train_pass("cpu", use_oracle=True)
    cpu = _TRAIN_CPU["r1"]
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    report = {"depth": rel(gpu["depth"], cpu["depth"]), "bev": rel(gpu["bev"], cpu["bev"]), "reg": rel(gpu["reg"], cpu["reg"])}
    grads = {"grad " + n: rel(gpu["grads"][n], cpu["grads"][n]) for n in GRAD_NAMES}
    print(policy, {k: "%.2e" % v for k, v in {**report, **grads}.items()})
    for k in cpu["losses"]:
        assert abs(gpu["losses"][k] - cpu["losses"][k]) <= 1e-3 * max(abs(cpu["losses"][k]), 1e-3), (k, gpu["losses"][k], cpu["losses"][k])
    assert abs(gpu["depth_loss"] - cpu["depth_loss"]) <= 1e-3 * abs(cpu["depth_loss"])
    bad = {k: v for k, v in report.items() if not v <= 1e-3}
    assert not bad, bad
    assert max(grads.values()) <= 1e-1, grads          # sanity only: see the docstring


def test_camera_only_config1_trains_and_detects_at_full_size(cuda):
    """BASELINE.json configs[1] (the reference's camera-only stage-1 config, cam_stream/LSS.py:30-123: torch SyncBN everywhere,
    lc_fusion=False, head on the 256-channel camera BEV) on the GPU: two bf16 training steps at R1 with the reference's four
    loss keys, then the test-time path (decode + rotated NMS) on one frame.  SyncBN in training mode needs a process group:
    a one-rank RCCL group in a child process, as the reference's own benchmark tool sets one up
    (tools/analysis_tools/benchmark.py:16-18)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, math, torch, torch.distributed as dist\n"
        f"sys.path[:0] = [{root!r}, {os.path.join(root, 'omnihd-scenes_amd')!r}]\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "from omnihd_amd.harness import FusionTrainStep, seed_miopen_db\n"
        "seed_miopen_db()\n"
        "st = FusionTrainStep(res='r1', batch=1, radar_dims=7, device='cuda:0', dtype='bf16', sets=1, task='camera')\n"
        "m = st.raw_model\n"
        "assert not hasattr(m, 'reduc_conv') and m.pts_bbox_head.conv_cls.in_channels == 256\n"
        "assert any(isinstance(x, torch.nn.SyncBatchNorm) for x in m.img_backbone.modules())\n"
        "l = [float(st.step().detach()) for _ in range(2)]\n"
        "assert all(math.isfinite(v) for v in l), l\n"
        "assert set(st.last_losses) == {'loss_cls', 'loss_bbox', 'loss_dir', 'img_depth_loss'}, list(st.last_losses)\n"
        "m.eval()\n"
        "torch.nn.init.constant_(m.pts_bbox_head.conv_cls.bias, -2.0)\n"
        "b = st.batches[0]\n"
        "with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):\n"
        "    out = m(return_loss=False, rescale=True, points=[b['points']], img_metas=[b['img_metas']], img=[b['img']])\n"
        "boxes = out[0]['pts_bbox']['boxes_3d'].tensor\n"
        "assert len(out) == 1 and boxes.shape[-1] == 9 and len(boxes) <= 500, boxes.shape\n"
        "dist.destroy_process_group()\n"
        "print('CAMERA_OK', l, len(boxes))\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 90), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "CAMERA_OK" in out.stdout, out.stderr[-3000:]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_full_size_r2_training_step_runs_and_is_finite(cuda, dtype):
    """BASELINE configs[2] at the repo's own resolution (final_dim 544 x 960, 8 radar channels: bevfusion.py:28,53,164): two
    training steps of the reference config, finite losses with the reference's four loss keys, the R2 pooling plan
    (4.5 M points) built once and on the default kernels (direct forward, patch backward)."""
    from omnihd_amd.harness import FusionTrainStep
    from omnihd_amd import plan as P
    st = FusionTrainStep(res="r2", batch=1, radar_dims=8, device="cuda:0", dtype=dtype, sets=1)
    P.TIMING = []
    try:
        losses = [float(st.step().detach()) for _ in range(2)]
        kinds = [k for k, _, _ in P.TIMING]
    finally:
        P.TIMING = None
    torch.cuda.synchronize()
    assert np.isfinite(losses).all(), losses
    assert set(st.last_losses) == {"loss_cls", "loss_bbox", "loss_dir", "img_depth_loss"}
    lss = st.raw_model.lift_splat_shot_vis
    assert len(lss._plans) == 1
    plan = next(iter(lss._plans.values()))
    assert plan.n_points == 4503872 and plan.feat_hw == 136 * 240 and plan.layout == "byxz"
    assert kinds.count("fwd") == 2 and kinds.count("bwd") == 2              # direct forward and patch backward ran
    assert kinds.count("plan") == 1                                          # ... on ONE plan, built on the device (round 6)
    from omnihd_amd.pool_plan import DevicePoolPlan
    assert isinstance(plan, DevicePoolPlan) and plan.counts()["tiles"] > 6000


def test_full_size_detector_bf16_step_runs_and_is_finite(cuda):
    """One R1 training step of the reference config (bf16 autocast, channels-last): finite losses with the
    reference's four loss keys; plan cache hit on the second step."""
    from omnihd_amd.harness import FusionTrainStep
    st = FusionTrainStep(res="r1", batch=1, radar_dims=7, device="cuda:0", dtype="bf16", sets=1)
    l0 = float(st.step().detach())
    lss = st.raw_model.lift_splat_shot_vis
    assert len(lss._plans) == 1
    plan = next(iter(lss._plans.values()))
    assert plan.n_points > 1_900_000 and plan.layout == "byxz"
    l1 = float(st.step().detach())
    assert len(lss._plans) == 1
    assert np.isfinite([l0, l1]).all()
    assert set(st.last_losses) == {"loss_cls", "loss_bbox", "loss_dir", "img_depth_loss"}


@pytest.mark.parametrize("policy", CONV_POLICIES)
def test_occupancy_variant_tiny_parity_and_full_size_step(cuda, policy, monkeypatch):
    """BEVF_FasterRCNN_MTL with the occupancy head — tiny model GPU (HIP ops) vs CPU (oracle ops)
    in fp32, then one full-size bf16 step of the reference's occupancy configuration.
    The convolution policy is PINNED like in the sibling tests (round 6): under the default measured choice the kernel of a
    geometry seen for the first time is picked by a timing race, so the last bits — and with them single ReLU masks of these
    few-pixel maps, see ``_grads_agree`` — followed the box's timing (the test failed 2 of 7 runs once the plan of a new
    calibration no longer synchronised the device: profiles/round6/occ_flaky.txt)."""
    import contextlib
    from omnihd_amd.harness import FusionTrainStep
    from oracle.torch_shim import oracle_ops
    monkeypatch.setenv("OMNIHD_FP32_CONV", policy)

    def run(device, use_oracle):
        with (oracle_ops() if use_oracle else contextlib.nullcontext()):
            st = FusionTrainStep(res="tiny", batch=2, radar_dims=7, device=device, seed=4, dtype="fp32", channels_last=False,
                                 sets=1, task="occ")
            m, b = st.raw_model, st.batches[0]
            m.eval()
            for mod in m.modules():
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
            losses = m(return_loss=True, **b)
            sum(v for v in losses.values()).backward()
            g = m.pts_bbox_head.task_decoders["occ"].final_conv.conv.weight.grad.detach().cpu()
            return {k: float(v) for k, v in losses.items()}, g
    gl, gg = run("cuda:0", False)
    cl, cg = run("cpu", True)
    assert set(gl) == {"loss_ssc", "loss_occ", "occ_sum", "img_depth_loss"}
    for k in cl:
        assert abs(gl[k] - cl[k]) <= 1e-3 * max(abs(cl[k]), 1e-3), (k, gl[k], cl[k])
    _grads_agree({"grads": {"g": gg}}, {"grads": {"g": cg}}, ["g"], policy)
    if policy != "split":
        return                                   # the full-size step once
    st = FusionTrainStep(res="r1", batch=1, radar_dims=7, device="cuda:0", dtype="bf16", task="occ")
    l0 = float(st.step().detach())
    l1 = float(st.step().detach())
    assert np.isfinite(l0) and np.isfinite(l1)
    assert set(st.last_losses) == {"loss_ssc", "loss_occ", "occ_sum", "img_depth_loss"}


def test_ddp_wrapped_step_on_one_gpu(cuda):
    """DistributedDataParallel over RCCL with a single rank: the reducer's hooks, bucket views and our autograd functions
    (weight-gradient chain, fused BatchNorm, pooling) in one bf16 training step; run in a child process so that the
    process group does not leak into the other tests."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, math, torch, torch.distributed as dist\n"
        f"sys.path[:0] = [{root!r}, {os.path.join(root, 'omnihd-scenes_amd')!r}]\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "from omnihd_amd.harness import FusionTrainStep\n"
        "st = FusionTrainStep(res='r1', batch=1, radar_dims=7, device='cuda:0', dtype='bf16', ddp=True)\n"
        "l = [float(st.step().detach()) for _ in range(3)]\n"
        "torch.cuda.synchronize()\n"
        "missing = [n for n, p in st.raw_model.named_parameters() if p.requires_grad and p.grad is None]\n"
        "assert all(math.isfinite(v) for v in l) and not missing, (l, missing[:5])\n"
        "dist.destroy_process_group()\n"
        "print('DDP_OK', l)\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29400 + os.getpid() % 500), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "DDP_OK" in out.stdout, out.stderr[-2000:]


def test_bf16_step_deviation_from_the_fp32_step(cuda):
    """The reference trains in fp32; the bf16-autocast step is the fast variant.  Same weights (seeded init), same R1
    frame, train-mode BatchNorm in both: the fused BEV feature, the box-regression map and the losses of the bf16 step
    stay within the stated bounds of the fp32 step (measured: see DESIGN.md section 5; north_star's 1e-3 is an fp32
    claim, checked against the oracle in test_tiny_detector_hip_ops_match_oracle_ops).  The rank tables and the voxel
    assignment are dtype-independent and must be identical."""
    from omnihd_amd.harness import FusionTrainStep
    outs = {}
    for dt in ("fp32", "bf16"):
        st = FusionTrainStep(res="r1", batch=1, radar_dims=7, device="cuda:0", dtype=dt, sets=1, seed=77)
        m, b = st.raw_model, st.batches[0]
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=dt == "bf16"):
            fd = m.extract_feat(b["points"], img=b["img"], img_metas=b["img_metas"])
            cls, reg, dirc = m.pts_bbox_head(fd["pts_feats"])
            losses = m.pts_bbox_head.loss(cls, reg, dirc, b["gt_bboxes_3d"], b["gt_labels_3d"], b["img_metas"])
            depth_loss, _ = m.lift_splat_shot_vis.get_depth_loss(b["img_depth"], fd["depth_dist"], "kld")
        plan = next(iter(m.lift_splat_shot_vis._plans.values()))     # built on the device: point words (ranks_depth | closing) + row CSR
        outs[dt] = dict(bev=fd["pts_feats"][0].float().cpu(), reg=reg[0].float().cpu(), cls=cls[0].float().cpu(),
                        depth=fd["depth_dist"].float().cpu(), ranks=plan.pt[:plan.n_points].cpu(), rows=plan.row_ptr.cpu(),
                        losses={k: float(v[0]) for k, v in losses.items()}, depth_loss=float(depth_loss))
        del st, m
        torch.cuda.empty_cache()
    a, b_ = outs["fp32"], outs["bf16"]
    assert torch.equal(a["ranks"], b_["ranks"]) and torch.equal(a["rows"], b_["rows"])      # indexing never depends on the dtype
    nrm = lambda x, y: float((x - y).norm() / (y.norm() + 1e-12))
    dev = {k: nrm(b_[k], a[k]) for k in ("bev", "reg", "cls", "depth")}
    dev.update({k: abs(b_["losses"][k] - a["losses"][k]) / max(abs(a["losses"][k]), 1e-6) for k in a["losses"]})
    dev["depth_loss"] = abs(b_["depth_loss"] - a["depth_loss"]) / abs(a["depth_loss"])
    print("bf16 vs fp32 deviation (relative L2 / relative loss):", {k: round(v, 5) for k, v in dev.items()})
    # measured on MI355X (round 2, random-init weights): bev 6.7e-2, reg 6.8e-2, depth 6.8e-2, cls 2.2e-3 relative L2;
    # losses 1e-5 .. 1.2e-3 relative.  ~50 bf16 convolutions deep, each rounding its output to 8 mantissa bits.
    assert dev["bev"] < 1e-1 and dev["depth"] < 1e-1 and dev["reg"] < 1e-1, dev
    assert dev["cls"] < 1e-2, dev
    assert all(dev[k] < 5e-3 for k in list(a["losses"]) + ["depth_loss"]), dev

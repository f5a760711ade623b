"""GPU parity of the TF32-grade form of the fp32 step's convolutions (round 6, OMNIHD_FP32_CONV=f16; csrc/conv_igemm.hip and
csrc/conv_wgrad_nhwc.hip with F16 = true): ONE IEEE-half MFMA product per fp32 product, fp32 accumulation — the precision the
reference trains at (TF32 is left on: tools/train.py:150-153; 11 significant bits per operand either way).

Two bars per kernel:
  * against torch's fp32 convolution of the SAME half-rounded operands: only the summation order differs, <= 2e-5 of the
    largest reference value (the kernel is exact about what it was given);
  * against torch's fp32 convolution of the fp32 operands: the rounding of the form itself, <= 1e-3 (operands rounded to
    2^-11 relative, errors averaging over the reduction).
Reference layers: BEV encoder cam_stream_lss_bevpoolv2_depthnet.py:201-214, fusion conv bevf_faster_rcnn_bevdepth.py:61-72."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("n", [1, 7, 8, 4099, 64 * 33 * 50])
def test_cast_is_round_to_nearest_and_the_scale_is_an_exact_power_of_two(cuda, n):
    from omnihd_amd import ops
    torch.manual_seed(n)
    x = torch.randn(n, device=cuda) * torch.logspace(-12, 3, n, device=cuda)
    h, inv = ops.cast_f16(x)
    assert inv is None and h.dtype == torch.float16 and torch.equal(h, x.half())
    for scale in (1e-9, 1.0, 3e7):
        xs = x * scale
        h, inv = ops.cast_f16(xs, scaled=True)
        inv_f = float(inv)
        m, e = torch.frexp(torch.tensor(inv_f))
        assert float(m) == 0.5, "the scale is a power of two"
        amax_scaled = float(xs.abs().max()) / inv_f
        assert 2.0 ** 14 <= amax_scaled < 2.0 ** 15, amax_scaled          # below the half range's 65504 with a bit to spare
        assert torch.equal(h, (xs / inv_f).half()) and bool(torch.isfinite(h).all())
    z, inv = ops.cast_f16(torch.zeros(n, device=cuda), scaled=True)
    assert float(inv) == 1.0 and not bool(z.any())


def test_unscaled_cast_saturates_finite_values_beyond_the_half_range(cuda):
    """TF32 keeps fp32's exponent range; the half form cannot — but a large finite activation must not turn into an infinity
    (and then NaNs) silently: it saturates.  Scaled casts never come near the range."""
    from omnihd_amd import ops
    inf, nan = float("inf"), float("nan")
    x = torch.tensor([7e4, -1e9, inf, -inf, nan, 65519.0, -65504.0, 1.0, 3e38, 0.0, -0.0, 6e-8, 1e-9], device=cuda)
    h, _ = ops.cast_f16(x)
    want = torch.tensor([65504.0, -65504.0, inf, -inf, nan, 65504.0, -65504.0, 1.0, 65504.0, 0.0, -0.0, 6e-8, 0.0]).half().to(cuda)
    assert torch.equal(h[:4], want[:4]) and bool(torch.isnan(h[4])) and torch.equal(h[5:], want[5:]), h
    g16, inv = ops.cast_f16(torch.tensor([3e38, -1e30, 1.0], device=cuda), scaled=True)
    assert bool(torch.isfinite(g16).all()) and float(g16[0].float() * inv) == pytest.approx(3e38, rel=1e-3)


def test_cast_keeps_the_memory_format(cuda):
    from omnihd_amd import ops
    x = _cl(torch.randn(2, 64, 6, 10, device=cuda))
    h, _ = ops.cast_f16(x)
    assert h.is_contiguous(memory_format=torch.channels_last) and torch.equal(h, x.half())


GEOMS = [  # B, H, W, cin, cout, k, dil, tile          (300: row-shift kernel, 256: 4x2 waves, 129: 2-stage 128 tile, 128: 4-stage)
    (1, 160, 240, 128, 256, 3, 1, 300), (1, 160, 240, 128, 256, 3, 1, 256), (1, 160, 240, 128, 256, 3, 1, 129),
    (1, 160, 240, 128, 256, 3, 1, 128), (1, 160, 240, 64, 64, 3, 1, 0), (2, 33, 50, 192, 136, 3, 2, 300),
    (6, 64, 176, 256, 256, 3, 6, 300), (1, 37, 41, 128, 72, 3, 12, 0), (2, 20, 30, 256, 128, 1, 1, 0),
    (1, 64, 176, 1280, 256, 1, 1, 0), (1, 9, 7, 64, 8, 3, 1, 0),
]


@pytest.mark.parametrize("B,H,W,cin,cout,k,dil,tile", GEOMS)
def test_half_forward_and_data_gradient(cuda, B, H, W, cin, cout, k, dil, tile):
    from omnihd_amd import ops
    torch.manual_seed(B * H + cin + k)
    x = _cl(torch.randn(B, cin, H, W, device=cuda))
    w = _cl(torch.randn(cout, cin, k, k, device=cuda) * (2.0 / (cin * k * k)) ** 0.5).requires_grad_()
    bias = torch.randn(cout, device=cuda)
    pad = dil * (k // 2)
    x16, _ = ops.cast_f16(x)
    w16 = ops.f16_weight(w)
    assert w16.shape == w.shape and torch.equal(w16, w.detach().half())
    got = ops.conv_fwd_f16(x16, w16, bias, None, dil, tile)
    assert got.dtype == torch.float32 and got.is_contiguous(memory_format=torch.channels_last)
    exact = F.conv2d(x16.float(), w16.float(), bias, padding=pad, dilation=dil)
    full = F.conv2d(x, w.detach(), bias, padding=pad, dilation=dil)
    assert _rel(got, exact) <= 2e-5, _rel(got, exact)
    assert _rel(got, full) <= 1e-3, _rel(got, full)
    assert torch.equal(got, ops.conv_fwd_f16(x16, w16, bias, None, dil, tile))            # deterministic
    if cout % 64 == 0:
        g = torch.randn_like(full) * 3e-6                       # gradient-sized values: far below the half range without the scale
        g16, inv = ops.cast_f16(g, scaled=True)
        wd = ops.f16_weight(w, dgrad=True)
        assert wd.shape == (cin, cout, k, k) and torch.equal(wd, _cl(w.detach().half().flip(2, 3).transpose(0, 1)))
        got_gx = ops.conv_fwd_f16(g16, wd, None, inv, dil, tile)
        exact_gx = torch.nn.grad.conv2d_input(x.shape, w16.float(), g16.float() * inv, padding=pad, dilation=dil)
        full_gx = torch.nn.grad.conv2d_input(x.shape, w.detach(), g, padding=pad, dilation=dil)
        assert _rel(got_gx, exact_gx) <= 2e-5, _rel(got_gx, exact_gx)
        assert _rel(got_gx, full_gx) <= 1e-3, _rel(got_gx, full_gx)


WGRAD_GEOMS = [  # B, H, W, cin, cout, k, dil        (three-taps kernel: 3x3 with cin <= 128; generic otherwise; one and several slabs)
    (1, 160, 240, 128, 256, 3, 1), (1, 160, 240, 64, 64, 3, 1), (6, 64, 176, 256, 256, 3, 1), (2, 20, 30, 256, 128, 1, 1),
    (1, 64, 176, 1280, 256, 1, 1), (2, 33, 50, 192, 128, 3, 2), (1, 12, 10, 512, 512, 3, 1), (6, 16, 44, 512, 512, 3, 1),
]


@pytest.mark.parametrize("B,H,W,cin,cout,k,dil", WGRAD_GEOMS)
def test_half_weight_gradient(cuda, B, H, W, cin, cout, k, dil):
    from omnihd_amd import ops
    pad = dil * (k // 2)
    if not ops.wgrad_nhwc_preferred(B, H, W, cin, H, W, cout, k, 1, pad, dil):
        pytest.skip("geometry left to the library by the NHWC weight-gradient kernel")
    torch.manual_seed(H + cin + k)
    x = _cl(torch.randn(B, cin, H, W, device=cuda))
    g = _cl(torch.randn(B, cout, H, W, device=cuda) * 2e-7)
    x16, _ = ops.cast_f16(x)
    g16, inv = ops.cast_f16(g, scaled=True)
    got = ops.conv_wgrad_f16(x16, g16, inv, k, 1, pad, dil)
    assert got.shape == (cout, cin, k, k) and got.dtype == torch.float32
    exact = torch.nn.grad.conv2d_weight(x16.double(), (cout, cin, k, k), g16.double() * float(inv), padding=pad, dilation=dil)
    full = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, k, k), g.double(), padding=pad, dilation=dil)
    assert _rel(got, exact) <= 2e-5, _rel(got, exact)
    assert _rel(got, full) <= 1e-3, _rel(got, full)
    assert torch.equal(got, ops.conv_wgrad_f16(x16, g16, inv, k, 1, pad, dil))


def test_module_path_under_the_policy(cuda, monkeypatch):
    """BevConv2d under OMNIHD_FP32_CONV=f16: stride-1 layers with 64-multiple channels run the half form in all three directions
    (1e-3 against nn.Conv2d in fp32); the strided layer stays on the fp32-grade kernels."""
    from omnihd_amd import ops
    from omnihd_amd.mm.bricks import use_bev_conv
    monkeypatch.setenv("OMNIHD_FP32_CONV", "f16")
    torch.manual_seed(2)
    mk = lambda: torch.nn.Sequential(torch.nn.Conv2d(64, 128, 3, padding=1, bias=False), torch.nn.Tanh(),
                                     torch.nn.Conv2d(128, 128, 3, stride=2, padding=1, bias=True), torch.nn.Tanh(),
                                     torch.nn.Conv2d(128, 64, 1, bias=True), torch.nn.Tanh(),
                                     torch.nn.Conv2d(64, 64, 3, padding=2, dilation=2, bias=False)).to(cuda).to(
                                         memory_format=torch.channels_last)
    ref, m = mk(), mk()
    m.load_state_dict(ref.state_dict())
    assert use_bev_conv(m) == 4
    x = _cl(torch.randn(2, 64, 24, 40, device=cuda))
    res = []
    for mod in (m, ref):
        mod.train()
        xi = x.clone().requires_grad_()
        y = mod(xi)
        y.square().mean().backward()
        res.append([y.detach(), xi.grad] + [p.grad for p in mod.parameters()])
    assert "ConvF16" in type(m(x).grad_fn).__name__
    assert "ConvF16" in type(m[0](x).grad_fn).__name__ and "ConvF16" not in type(m[2](m[0](x)).grad_fn).__name__
    for a, b in zip(*res):
        assert a.dtype == torch.float32 and _rel(a, b) <= 1e-3, _rel(a, b)
    # the default policy is untouched by the form's existence
    monkeypatch.delenv("OMNIHD_FP32_CONV")
    assert "ConvF16" not in type(m(x).grad_fn).__name__


def test_temporary_weights_are_converted_directly_without_a_table_upload(cuda, monkeypatch):
    """A weight computed in the forward (the block-diagonal matrix DCN rebuilds every step) is never seen again under its id: it
    is converted directly — a cached image would be rebuilt every step through a table whose upload is a blocking copy (2.4 ms per
    step in the first host profile of the form, profiles/round6/host_f16.txt)."""
    from omnihd_amd import ops
    monkeypatch.setenv("OMNIHD_FP32_CONV", "f16")
    torch.manual_seed(6)
    w = (torch.randn(64, 128, 3, 3, device=cuda) * 0.05).requires_grad_()
    x = _cl(torch.randn(1, 128, 10, 14, device=cuda)).requires_grad_()
    uploads = ops.WIMG_STATS["miss"]
    y = ops.conv_split(x, w * 2.0, None, (1, 1), (1, 1))
    assert "ConvF16" in type(y.grad_fn).__name__
    y.square().mean().backward()
    assert ops.WIMG_STATS["miss"] == uploads
    x2, w2 = x.detach().clone().requires_grad_(), w.detach().clone().requires_grad_()
    F.conv2d(x2, w2 * 2.0, None, padding=1).square().mean().backward()
    assert _rel(x.grad, x2.grad) <= 1e-3 and _rel(w.grad, w2.grad) <= 1e-3, (_rel(x.grad, x2.grad), _rel(w.grad, w2.grad))


def test_producers_hand_over_the_half_plane_and_the_gradient_amax(cuda, monkeypatch):
    """Under the policy the BatchNorm / frozen-BatchNorm epilogues write the half plane of their output once a TF32-grade convolution
    has asked for it (no cast pass there), and their backward accumulates max |gx| for the scaled cast of the convolution in front of
    them (no amax pass there).  Both are exact replacements: the results are BIT-identical to the passes they replace
    (OMNIHD_F16_HANDOVER=0)."""
    from omnihd_amd import ops
    monkeypatch.setenv("OMNIHD_FP32_CONV", "f16")
    ops._HALF_WANTED.clear(); ops._PLANES_UNUSED.clear()
    torch.manual_seed(8)
    w1 = _cl(torch.randn(64, 64, 3, 3, device=cuda) * 0.05).requires_grad_()
    w2 = _cl(torch.randn(128, 64, 1, 1, device=cuda) * 0.1).requires_grad_()
    w3 = _cl(torch.randn(64, 128, 3, 3, device=cuda) * 0.03).requires_grad_()
    gamma, beta = (torch.rand(64, device=cuda) + 0.5).requires_grad_(), (torch.randn(64, device=cuda) * 0.1).requires_grad_()
    scale, shift = torch.rand(128, device=cuda) + 0.5, torch.randn(128, device=cuda) * 0.1
    x0 = _cl(torch.randn(2, 64, 20, 28, device=cuda))

    def run():
        x = x0.clone().requires_grad_()
        y1 = ops.conv_split(x, w1, None, (1, 1), (1, 1))
        y2 = ops.bn_train_act(y1, gamma, beta, torch.zeros(64, device=cuda), torch.ones(64, device=cuda), 0.1, 1e-5, relu=True)
        y3 = ops.conv_split(y2, w2, None, (1, 1), (0, 0))
        y4 = ops.affine_act(y3, scale, shift, None, True)
        y5 = ops.conv_split(y4, w3, None, (1, 1), (1, 1))
        grads = torch.autograd.grad(y5.square().mean(), [x, w1, w2, w3, gamma, beta])
        return [y5.detach()] + [g.detach() for g in grads]

    def counters():
        return ops.HANDOVER_STATS.get("taken_half", 0), ops.FAST_PATHS.get("f16_amax_from_producer", 0)

    monkeypatch.setenv("OMNIHD_F16_HANDOVER", "0")
    c0 = counters()
    plain = run()
    assert counters() == c0
    monkeypatch.setenv("OMNIHD_F16_HANDOVER", "1")
    run()                                                            # the convolutions ask; from now on the producers write half planes
    c0 = counters()
    handed = run()
    assert counters() == (c0[0] + 2, c0[1] + 2)
    for a, b in zip(plain, handed):
        assert torch.equal(a, b)
    # and against plain fp32 torch, to the grade of the form (gradients in relative L2: a ReLU whose input lies within the form's
    # rounding of zero flips its mask and moves single entries by their full size)
    x = x0.clone().requires_grad_()
    y = F.conv2d(x, w1, None, padding=1)
    y = F.relu(F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5))
    y = F.relu(F.conv2d(y, w2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    y = F.conv2d(y, w3, None, padding=1)
    want = [y.detach()] + list(torch.autograd.grad(y.square().mean(), [x, w1, w2, w3, gamma, beta]))
    assert _rel(handed[0], want[0]) <= 4e-3, _rel(handed[0], want[0])
    for a, b in zip(handed[1:], want[1:]):
        l2 = float((a.double() - b.double()).norm() / b.double().norm())
        assert l2 <= 3e-2, l2


def test_half_images_follow_the_optimiser_with_one_launch(cuda, monkeypatch):
    from omnihd_amd import ops
    monkeypatch.setenv("OMNIHD_FP32_CONV", "f16")
    torch.manual_seed(4)
    ws = [_cl(torch.randn(co, ci, k, k, device=cuda) * 0.05).requires_grad_() for co, ci, k in ((64, 64, 3), (128, 64, 1), (192, 128, 3))]
    x = _cl(torch.randn(1, 64, 12, 20, device=cuda)).requires_grad_()
    y = ops.conv_split(x, ws[0], None, (1, 1), (1, 1))
    y = ops.conv_split(y, ws[1], None, (1, 1), (0, 0))
    y = ops.conv_split(y, ws[2], None, (1, 1), (1, 1))
    y.square().mean().backward()
    opt = torch.optim.SGD(ws, lr=0.5)
    opt.step()
    assert ops.refresh_f16_shadows() == 3 and ops.refresh_f16_shadows() == 0
    for w in ws:
        assert torch.equal(ops.f16_weight(w), w.detach().half())
        assert torch.equal(ops.f16_weight(w, dgrad=True), _cl(w.detach().half().flip(2, 3).transpose(0, 1)))


def _round_tf32(t):
    """fp32 -> nearest value with 10 explicit mantissa bits (ties to even), as an fp32 tensor: what a TF32 matrix unit reads."""
    b = t.detach().clone().view(torch.int32)
    b += 0xFFF + ((b >> 13) & 1)
    b &= ~0x1FFF
    return b.view(torch.float32)


@pytest.mark.parametrize("res", ["r1", "r2"])
def test_full_size_forward_deviates_from_the_fp32_grade_run_as_tf32_operand_rounding_does(cuda, res, monkeypatch):
    """VERDICT round 5 #4 asked for tests/test_detector_gpu.py:131's gate (fused BEV feature and box regressions <= 1e-3 at R1 and
    R2) under the TF32-grade form.  Measured, it is NOT met — and cannot be by ANY arithmetic with 11-bit operands: the same
    forward with every convolution operand rounded to TF32 and the products then taken at fp32 grade (what the reference's cuDNN
    does with allow_tf32, tools/train.py:150-153) deviates from the fp32-grade run by the same amount.  One forward of the reference
    config (BatchNorm in inference mode, seeded weights) three times: fp32-grade split kernels, the half form, the TF32 emulation.
    Asserted: the half form is within 1.5 x the TF32 emulation's deviation (relative L2, BEV feature and box regressions), and
    both within 1e-2; printed: all numbers.  Measured (profiles/round6/pytest_f16_gates.txt), half form / TF32 emulation:
    R1 BEV 2.78e-3 / 2.78e-3, box regressions 2.95e-3 / 2.97e-3; R2 BEV 6.02e-3 / 5.69e-3, box regressions 6.47e-3 / 6.11e-3."""
    from omnihd_amd import ops
    from omnihd_amd.harness import FusionTrainStep
    out = {}
    real_conv_split = ops.conv_split
    for policy in ("split", "f16", "tf32emu"):
        monkeypatch.setenv("OMNIHD_FP32_CONV", "split" if policy == "tf32emu" else policy)
        if policy == "tf32emu":
            monkeypatch.setattr(ops, "conv_split", lambda x, w, b, *a, **k: real_conv_split(_round_tf32(x), _round_tf32(w), b, *a, **k))
        st = FusionTrainStep(res=res, batch=1, radar_dims=7 if res == "r1" else 8, device="cuda:0", seed=5, dtype="fp32",
                             channels_last=True, sets=1)
        m, b = st.raw_model, st.batches[0]
        m.eval()
        with torch.no_grad():
            fd = m.extract_feat(b["points"], img=b["img"], img_metas=b["img_metas"])
            cls, reg, _ = m.pts_bbox_head(fd["pts_feats"])
        out[policy] = dict(bev=fd["pts_feats"][0].float(), cls=cls[0].float(), reg=reg[0].float())
        del st, m, fd
    rel = lambda a, b: float((a - b).norm() / b.norm())
    mx = lambda a, b: float((a - b).abs().max() / b.abs().max())
    report = {p: {k: rel(out[p][k], out["split"][k]) for k in ("bev", "cls", "reg")} for p in ("f16", "tf32emu")}
    report_max = {p: {k: mx(out[p][k], out["split"][k]) for k in ("bev", "cls", "reg")} for p in ("f16", "tf32emu")}
    print("\n", res, "relative L2 against the fp32-grade run:", {p: {k: "%.2e" % v for k, v in r.items()} for p, r in report.items()})
    print(res, "max |diff| / max |ref|:", {p: {k: "%.2e" % v for k, v in r.items()} for p, r in report_max.items()})
    for k in ("bev", "reg"):
        assert 0.0 < report["f16"][k] <= 1.5 * report["tf32emu"][k], (k, report)
        assert report["f16"][k] <= 1e-2 and report["tf32emu"][k] <= 1e-2, (k, report)


def test_training_steps_under_the_policy_track_the_fp32_grade_steps(cuda, monkeypatch):
    """Four optimiser steps of the R1 step under both policies from the same seed: finite, and the loss trajectories agree to
    the grade of the form (printed)."""
    from omnihd_amd.harness import FusionTrainStep
    traj = {}
    for policy in ("split", "f16"):
        monkeypatch.setenv("OMNIHD_FP32_CONV", policy)
        st = FusionTrainStep(res="r1", batch=1, radar_dims=7, device="cuda:0", seed=5, dtype="fp32", channels_last=True, sets=1)
        traj[policy] = [float(st.step()) for _ in range(4)]
        del st
    print("loss trajectories:", traj)
    assert all(v == v and abs(v) < 1e6 for v in traj["f16"])
    assert abs(traj["f16"][0] - traj["split"][0]) <= 2e-3 * abs(traj["split"][0]), traj

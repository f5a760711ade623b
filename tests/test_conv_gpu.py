"""GPU parity of the hand-written MFMA weight-gradient kernel against torch's fp32 convolution backward
on the same bf16-rounded inputs (only the accumulation order differs: tolerance 2e-3 of the gradient's
max magnitude, bf16 products are exact in fp32)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def ref_wgrad(x, g):
    w = torch.zeros(g.shape[1], x.shape[1], 3, 3, device=x.device, requires_grad=True)
    y = F.conv2d(x.float(), w, padding=1)
    return torch.autograd.grad(y, w, g.float())[0]


@pytest.mark.parametrize("B,H,W,cin,cout", [(1, 16, 24, 128, 128), (2, 9, 40, 256, 128), (1, 160, 240, 128, 256), (1, 7, 8, 384, 640),
                                             (2, 48, 47, 512, 256), (1, 160, 240, 640, 384), (3, 33, 50, 392, 264)])   # the last three: register-shifted taps (k_wgrad_shift)
def test_wgrad_matches_fp32_reference(cuda, B, H, W, cin, cout):
    from omnihd_amd import ops
    torch.manual_seed(B * H + cin)
    x = torch.randn(B, cin, H, W, device=cuda).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    g = (torch.randn(B, cout, H, W, device=cuda) * 0.1).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    got = ops.conv3x3_wgrad(x, g)
    assert got.shape == (cout, cin, 3, 3) and got.dtype == torch.float32
    want = ref_wgrad(x, g)
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-3 * scale
    assert torch.equal(got, ops.conv3x3_wgrad(x, g))          # deterministic


def test_border_taps_see_zero_padding(cuda):
    """x = 1 everywhere, g = 1 everywhere: dW[tap] = number of pixels whose shifted neighbour is inside."""
    from omnihd_amd import ops
    B, H, W, C = 1, 8, 16, 128
    x = torch.ones(B, C, H, W, device=cuda, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    g = torch.ones(B, C, H, W, device=cuda, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dw = ops.conv3x3_wgrad(x, g)
    want = torch.tensor([[(H - abs(dy)) * (W - abs(dx)) for dx in (-1, 0, 1)] for dy in (-1, 0, 1)], dtype=torch.float32, device=cuda)
    assert torch.equal(dw[5, 77], want) and torch.equal(dw[127, 0], want)


def test_bev_conv_module_trains_like_nn_conv2d(cuda):
    from omnihd_amd.mm.bricks import BevConv2d, use_bev_conv
    torch.manual_seed(0)
    seq = torch.nn.Sequential(torch.nn.Conv2d(128, 256, 3, padding=1, bias=False), torch.nn.ReLU(),
                              torch.nn.Conv2d(256, 128, 3, padding=1, bias=False)).to(cuda).to(memory_format=torch.channels_last)
    ref = torch.nn.Sequential(torch.nn.Conv2d(128, 256, 3, padding=1, bias=False), torch.nn.ReLU(),
                              torch.nn.Conv2d(256, 128, 3, padding=1, bias=False)).to(cuda).to(memory_format=torch.channels_last)
    ref.load_state_dict(seq.state_dict())
    use_bev_conv(seq)
    assert isinstance(seq[0], BevConv2d) and list(seq.state_dict()) == list(ref.state_dict())
    x = torch.randn(2, 128, 20, 24, device=cuda).contiguous(memory_format=torch.channels_last)
    for m in (seq, ref):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            m(x).float().square().mean().backward()
    for a, b in zip(seq.parameters(), ref.parameters()):
        assert a.grad.dtype == torch.float32
        torch.testing.assert_close(a.grad, b.grad, rtol=3e-2, atol=3e-2 * float(b.grad.abs().max()))


def test_dcn_hip_sampling_matches_torch_formulation(cuda):
    """HIP deformable sampling (fwd, grad wrt input, grad wrt offsets) vs the embedding_bag formulation in fp32 on
    the same bf16-rounded inputs; offsets reach beyond one pixel and outside the image."""
    from omnihd_amd.mm.dcn import DeformConv2dPack
    torch.manual_seed(0)
    m = DeformConv2dPack(64, 64, 3, padding=1, groups=4).to(cuda)
    torch.nn.init.normal_(m.conv_offset.weight, std=0.05)
    torch.nn.init.normal_(m.conv_offset.bias, std=1.2)
    x = torch.randn(2, 64, 12, 20, device=cuda).to(torch.bfloat16).float().contiguous(memory_format=torch.channels_last).requires_grad_()
    off = m.conv_offset(x).detach().requires_grad_()
    a = m._hip_sample_and_gemm(x, off, torch.bfloat16)
    b = m._gather_and_gemm(x, off, torch.float32)
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) <= 2e-2 * scale
    g = torch.randn_like(b)
    ga = torch.autograd.grad(a, [x, off, m.weight], g, retain_graph=True)
    gb = torch.autograd.grad(b, [x, off, m.weight], g)
    for u, v, name in zip(ga, gb, ("x", "offset", "weight")):
        assert float((u - v).abs().max()) <= 3e-2 * float(v.abs().max()), name
    again = m._hip_sample_and_gemm(x, off, torch.bfloat16)
    assert torch.equal(a, again)


@pytest.mark.parametrize("c,H,W", [(64, 12, 20), (256, 16, 44), (32, 5, 7)])
def test_dcn_hip_sampling_fp32_matches_torch_formulation(cuda, c, H, W):
    """The fp32 form of the sampling kernels (the reference's arithmetic; what the fp32 training step runs) against the
    embedding_bag formulation in fp32: forward, grad wrt input, offsets and weight to fp32 rounding (1e-5 of the largest
    value; the offset gradient sums C products per sample in a different order: 5e-5)."""
    from omnihd_amd.mm.dcn import DeformConv2dPack
    torch.manual_seed(c)
    m = DeformConv2dPack(c, 64, 3, padding=1, groups=4).to(cuda)
    torch.nn.init.normal_(m.conv_offset.weight, std=0.05)
    torch.nn.init.normal_(m.conv_offset.bias, std=1.2)
    x = torch.randn(2, c, H, W, device=cuda).contiguous(memory_format=torch.channels_last).requires_grad_()
    off = m.conv_offset(x).detach().requires_grad_()
    prev = torch.backends.cuda.matmul.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = False
    try:
        a = m._hip_sample_and_gemm(x, off, torch.float32)
        b = m._gather_and_gemm(x, off, torch.float32)
        assert a.dtype == torch.float32
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
        g = torch.randn_like(b)
        ga = torch.autograd.grad(a, [x, off, m.weight], g, retain_graph=True)
        gb = torch.autograd.grad(b, [x, off, m.weight], g)
    finally:
        torch.backends.cuda.matmul.allow_tf32 = prev
    for u, v, name in zip(ga, gb, ("x", "offset", "weight")):
        assert float((u - v).abs().max()) <= 5e-5 * float(v.abs().max()), name
    y = m(x)                                           # the module takes the HIP path in fp32 too
    assert y.dtype == torch.float32 and float((y - m._gather_and_gemm(x, m.conv_offset(x), torch.float32)).abs().max()) \
        <= 1e-5 * float(y.abs().max())


@pytest.mark.parametrize("c,H,W", [(32, 9, 14), (64, 6, 11)])
def test_dcn_hip_path_matches_the_independent_c_oracle(cuda, c, H, W):
    """the HIP deformable-sampling kernels + GEMM (fp32 form) against oracle/dcn_oracle.c — an
    independent C restatement of mmcv 1.4.0's deformable_im2col (channel order (dy, dx) per tap, `> -1 / < H` border rule,
    corner-wise zero padding; its own known answers are in tests/test_oracle.py) — not against the product's torch formulation:
    forward and the gradients with respect to input, offsets and weight; offsets up to several pixels, reaching outside the
    image.  1e-5 of the largest value (5e-5 for the offset gradient: C products per sample summed in another order)."""
    from omnihd_amd.mm.dcn import DeformConv2dPack
    from oracle import cpu as OC
    torch.manual_seed(c + H)
    m = DeformConv2dPack(c, 32, 3, padding=1, groups=4).to(cuda)
    x = torch.randn(2, c, H, W, device=cuda).contiguous(memory_format=torch.channels_last).requires_grad_()
    off = (torch.randn(2, 18, H, W, device=cuda) * 1.5 + 0.01).requires_grad_()
    off.data[0, :, 0, 0] = -3.0                                  # far outside: zero sample, zero gradients
    g = torch.randn(2, 32, H, W, device=cuda)
    prev = torch.backends.cuda.matmul.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = False
    try:
        y = m._hip_sample_and_gemm(x, off, torch.float32)
        gx, go, gw = torch.autograd.grad(y, [x, off, m.weight], g)
    finally:
        torch.backends.cuda.matmul.allow_tf32 = prev
    want = OC.deform_conv(x.detach().cpu().numpy(), off.detach().cpu().numpy(), m.weight.detach().cpu().numpy(), 1, 1, 1, groups=4,
                          grad_out=g.cpu().numpy())
    for got, w, name, tol in zip((y, gx, go, gw), want, ("output", "grad input", "grad offset", "grad weight"), (1e-5, 1e-5, 5e-5, 1e-5)):
        w = torch.from_numpy(w)
        assert got.shape == w.shape, name
        assert float((got.detach().cpu() - w).abs().max()) <= tol * float(w.abs().max()), name


@pytest.mark.parametrize("B,H,W,cin,cout,stride", [(2, 8, 22, 512, 128, 1), (6, 16, 44, 256, 1024, 1), (1, 9, 13, 128, 256, 2), (3, 7, 5, 2048, 512, 1)])
def test_wgrad_1x1_matches_fp32_reference(cuda, B, H, W, cin, cout, stride):
    from omnihd_amd import ops
    torch.manual_seed(cin + H)
    x = torch.randn(B, cin, H, W, device=cuda).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 1, 1, device=cuda) * 0.05).to(torch.bfloat16).requires_grad_()
    y = ops.conv_hip_wgrad(x, w, None, (stride, stride), (0, 0))
    g = (torch.randn_like(y.float()) * 0.1).to(torch.bfloat16)
    (gw,) = torch.autograd.grad(y, w, g)
    wr = w.detach().float().requires_grad_()
    yr = F.conv2d(x.float(), wr, None, stride)
    (want,) = torch.autograd.grad(yr, wr, g.float())
    assert gw.shape == want.shape
    assert float((gw.float() - want).abs().max()) <= 1e-2 * float(want.abs().max())      # bf16 output rounding


# ---------------------------------------------------------------------------------------------
# Frozen-BatchNorm epilogue (csrc/affine_act.hip)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,relu,with_res", [((6, 256, 16, 44), True, True), ((2, 64, 9, 7), True, False),
                                                 ((1, 8, 3, 5), False, True), ((3, 2048, 8, 22), False, False)])
def test_affine_act_forward_backward_match_torch(cuda, shape, relu, with_res):
    from omnihd_amd import ops
    g = torch.Generator(device="cpu").manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).to(cuda).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_()
    res = (torch.randn(shape, generator=g).to(cuda).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_()
           if with_res else None)
    scale = (torch.rand(shape[1], generator=g) + 0.5).to(cuda)
    shift = torch.randn(shape[1], generator=g).to(cuda)
    gy = torch.randn(shape, generator=g).to(cuda).bfloat16()          # NCHW-contiguous on purpose
    y = ops.affine_act(x, scale, shift, res, relu)
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    ref = x.detach().float() * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if with_res:
        ref = ref + res.detach().float()
    if relu:
        ref = ref.clamp_min(0)
    # one bf16 rounding of an fp32 value that may differ in its last fp32 bit (fma vs mul+add)
    torch.testing.assert_close(y.float(), ref.bfloat16().float(), rtol=2 ** -7, atol=1e-6)
    assert (y.float() - ref).abs().max() <= ref.abs().max() * 2 ** -8 + 1e-6
    y.backward(gy)
    mask = (y.detach().float() > 0) if relu else torch.ones_like(ref, dtype=torch.bool)
    want_res = (gy.float() * mask).bfloat16()
    want_x = (want_res.float() * scale.view(1, -1, 1, 1)).bfloat16()
    assert torch.equal(x.grad, want_x) and x.grad.is_contiguous(memory_format=torch.channels_last)
    if with_res:
        assert torch.equal(res.grad, want_res)


def test_frozen_bn_blocks_fused_path_matches_torch_composition(cuda, monkeypatch):
    """A ResNet stage with frozen BatchNorm under bf16 autocast: fused epilogues vs plain torch ops."""
    from omnihd_amd import ops
    from omnihd_amd.mm.resnet import ResNet
    torch.manual_seed(0)
    net = ResNet(depth=50, num_stages=2, strides=(1, 2), dilations=(1, 1), out_indices=(0, 1), frozen_stages=0,
                 norm_cfg=dict(type="BN", requires_grad=False), norm_eval=True).to(cuda).to(memory_format=torch.channels_last)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
    net.train()
    x = torch.randn(2, 3, 64, 96, device=cuda).contiguous(memory_format=torch.channels_last)

    def run(autocast=True):
        for p in net.parameters():
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            outs = net(x)
        sum(o.float().square().mean() for o in outs).backward()
        return [o.detach().float() for o in outs], {n: p.grad.detach().float().clone() for n, p in net.named_parameters()
                                                    if p.grad is not None}
    calls = []
    real = ops.affine_act
    monkeypatch.setattr(ops, "affine_act", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    fused_out, fused_grad = run()
    assert len(calls) >= 20                                    # every BN of the two stages took the fused path
    monkeypatch.setattr(ops, "affine_act_supported", lambda *a, **k: False)
    plain_out, plain_grad = run()
    exact_out, exact_grad = run(autocast=False)                # fp32 end to end: the yardstick for both bf16 runs
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-12))
    for f, p, e in zip(fused_out, plain_out, exact_out):
        assert f.shape == e.shape and rel(p, e) < 0.05
        assert rel(f, e) <= 1.5 * rel(p, e) + 1e-3, (rel(f, e), rel(p, e))
    assert fused_grad.keys() == plain_grad.keys() == exact_grad.keys() and len(fused_grad) > 10
    for k in fused_grad:
        assert rel(fused_grad[k], exact_grad[k]) <= 1.5 * rel(plain_grad[k], exact_grad[k]) + 5e-3, k


# ---------------------------------------------------------------------------------------------
# Generic convolution weight gradient (any 1x1 / 3x3 geometry of the detector) and transposed conv
# ---------------------------------------------------------------------------------------------
GEOMS = [  # B, H, W, cin, cout, k, stride, pad, dil
    (6, 16, 44, 256, 256, 3, 1, 1, 1),       # ResNet layer3 / FPN level: width not a multiple of 8
    (6, 8, 22, 512, 512, 3, 1, 1, 1),        # ResNet layer4
    (1, 40, 60, 256, 256, 3, 1, 1, 1),       # SECOND stage 3
    (1, 32, 48, 64, 64, 3, 1, 1, 1),         # SECOND stage 1: 64 channels (padded to 128 inside)
    (2, 24, 40, 256, 256, 3, 1, 6, 6),       # ASPP dilation 6
    (1, 40, 48, 128, 256, 3, 1, 18, 18),     # ASPP dilation 18 (> feature height/2)
    (2, 32, 48, 128, 128, 3, 2, 1, 1),       # ResNet stage entry, stride 2
    (1, 33, 47, 64, 128, 3, 2, 1, 1),        # odd input size, stride 2 (SECOND entry)
    (1, 20, 24, 384, 72, 1, 1, 0, 1),        # head regression conv: Cout 72
    (1, 20, 24, 384, 16, 1, 1, 0, 1),        # head direction conv
    (2, 16, 24, 256, 512, 1, 2, 0, 1),       # ResNet downsample 1x1 stride 2
    (1, 9, 10, 8, 8, 3, 1, 1, 1),            # smallest channel count
]


@pytest.mark.parametrize("B,H,W,cin,cout,k,stride,pad,dil", GEOMS)
def test_generic_wgrad_matches_fp32_reference(cuda, B, H, W, cin, cout, k, stride, pad, dil):
    from omnihd_amd import ops
    torch.manual_seed(H * W + cin + k)
    x = torch.randn(B, cin, H, W, device=cuda).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w0 = torch.zeros(cout, cin, k, k, device=cuda, requires_grad=True)
    y = F.conv2d(x.float(), w0, None, stride, pad, dil)
    g = (torch.randn_like(y) * 0.1).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    want = torch.autograd.grad(y, w0, g.float())[0]
    assert ops.conv_wgrad_supported(x, w0, (stride, stride), (pad, pad), (dil, dil))
    got = ops.conv_wgrad(x, g, k, stride, pad, dil)
    assert got.shape == want.shape and got.dtype == torch.float32
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-3 * scale, float((got - want).abs().max()) / scale
    assert torch.equal(got, ops.conv_wgrad(x, g, k, stride, pad, dil))          # deterministic


def test_bevconv2d_module_gradients_match_plain_conv(cuda):
    """BevConv2d inside autocast (dilated, strided, bias) vs nn.Conv2d: same forward, same three gradients."""
    from omnihd_amd.mm.bricks import BevConv2d
    torch.manual_seed(1)
    for kw in (dict(kernel_size=3, stride=2, padding=1, bias=True), dict(kernel_size=3, padding=12, dilation=12, bias=False),
               dict(kernel_size=1, bias=True)):
        ref = torch.nn.Conv2d(64, 72, **kw).to(cuda).to(memory_format=torch.channels_last)
        mod = BevConv2d(64, 72, **kw).to(cuda).to(memory_format=torch.channels_last)
        mod.load_state_dict(ref.state_dict())
        x1 = torch.randn(2, 64, 30, 44, device=cuda).contiguous(memory_format=torch.channels_last).requires_grad_()
        x2 = x1.detach().clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y1, y2 = ref(x1), mod(x2)
        # the forward runs on whichever implementation measured faster for the geometry (MIOpen: identical bits; the
        # implicit-GEMM kernel: fp32 accumulation in another order, one bf16 rounding)
        assert float((y1.float() - y2.float()).abs().max()) <= 2 ** -7 * float(y1.float().abs().max())
        gy = torch.randn_like(y1)
        y1.backward(gy); y2.backward(gy)
        torch.testing.assert_close(x2.grad, x1.grad, rtol=2e-2, atol=2e-2)
        assert mod.weight.grad.dtype == torch.float32
        err = float((mod.weight.grad - ref.weight.grad).abs().max()) / float(ref.weight.grad.abs().max())
        assert err < 1e-2, (kw, err)                                           # MIOpen's own wrw is bf16-accumulate-limited
        if kw.get("bias"):
            torch.testing.assert_close(mod.bias.grad, ref.bias.grad, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("cin,cout,k,H,W", [(64, 128, 1, 40, 60), (128, 128, 2, 20, 30), (256, 128, 4, 10, 15)])
def test_transposed_conv_weight_gradient(cuda, cin, cout, k, H, W):
    from omnihd_amd import ops
    torch.manual_seed(k)
    x = torch.randn(1, cin, H, W, device=cuda).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
    w = (torch.randn(cin, cout, k, k, device=cuda) * 0.05).requires_grad_()
    assert ops.deconv_supported(x, w, (k, k), (k, k), (0, 0), (0, 0), 1, (1, 1), None)
    y = ops.deconv_hip_wgrad(x, w, k)
    g = (torch.randn_like(y.float()) * 0.1).to(torch.bfloat16)
    y.backward(g)
    xr = x.detach().float().requires_grad_()
    wr = w.detach().to(torch.bfloat16).float().requires_grad_()
    yr = F.conv_transpose2d(xr, wr, None, stride=k)
    yr.backward(g.float())
    torch.testing.assert_close(y.float(), yr, rtol=2e-2, atol=2e-2)
    assert float((w.grad - wr.grad).abs().max()) <= 2e-3 * float(wr.grad.abs().max())
    assert float((x.grad.float() - xr.grad).abs().max()) <= 2e-2 * float(xr.grad.abs().max())


@pytest.mark.parametrize("B,H,W,cin,cout,k,dil,tile", [(1, 160, 240, 128, 256, 3, 1, 0), (1, 160, 240, 128, 256, 3, 1, 128),
                                                          (2, 20, 30, 64, 72, 3, 1, 0), (1, 33, 47, 192, 136, 3, 2, 128),
                                                          (1, 33, 47, 192, 136, 3, 2, 256), (3, 16, 44, 256, 64, 1, 1, 0),
                                                          (1, 64, 176, 256, 256, 3, 6, 256), (1, 160, 240, 128, 256, 3, 1, 300),
                                                          (2, 20, 30, 64, 72, 3, 1, 300), (1, 33, 47, 192, 136, 3, 2, 300),
                                                          (1, 64, 176, 256, 256, 3, 6, 300), (3, 17, 9, 64, 64, 3, 8, 300)])
def test_igemm_conv_forward_and_data_gradient_match_fp32_reference(cuda, B, H, W, cin, cout, k, dil, tile):
    """csrc/conv_igemm.hip against torch.nn.functional.conv2d in fp32 on the same bf16-rounded operands: forward with bias,
    and the data gradient as the same kernel on mirrored / transposed weights.  Bound: one bf16 rounding of the result
    (2^-8 relative) on top of fp32 accumulation in a different order."""
    from omnihd_amd import ops
    g = torch.Generator(device="cpu").manual_seed(B * H + cin + cout + k)
    x = torch.randn(B, cin, H, W, generator=g).to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(cuda).bfloat16()
    w = w.contiguous(memory_format=torch.channels_last)
    bias = torch.randn(cout, generator=g).to(cuda)
    gy = torch.randn(B, cout, H, W, generator=g).to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    assert ops.conv_fwd_supported(x.shape, cout, k, 1, dil * (k // 2), dil)
    y = ops.conv_fwd(x, w, bias, dilation=dil, tile=tile)
    xr = x.float().requires_grad_()
    yr = torch.nn.functional.conv2d(xr, w.float(), bias, stride=1, padding=dil * (k // 2), dilation=dil)
    assert y.shape == yr.shape and y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    err = (y.float() - yr).abs()
    assert float(err.max()) <= 2 ** -7 * float(yr.abs().max()), float(err.max() / yr.abs().max())
    assert float((y.float() - yr).norm() / yr.norm()) < 3e-3
    assert torch.equal(y, ops.conv_fwd(x, w, bias, dilation=dil, tile=tile))                    # deterministic
    if cout % 64 == 0:
        wt = ops.conv_dgrad_weights(w)
        assert wt.shape == (cin, cout, k, k)
        gx = ops.conv_fwd(gy, wt, None, dilation=dil, tile=tile)
        yr.backward(gy.float())
        assert float((gx.float() - xr.grad).norm() / xr.grad.norm()) < 3e-3
        assert float((gx.float() - xr.grad).abs().max()) <= 2 ** -7 * float(xr.grad.abs().max())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,c", [(67584, 59), (4099, 123), (513, 7), (2000, 300), (3, 1)])
def test_column_sums_any_width(cuda, dtype, rows, c):
    """omnihd_column_sums (bias gradient for channel counts that are not multiples of 8) against the float64 sums."""
    from omnihd_amd import ops
    a = torch.randn(rows, c, device=cuda, generator=torch.Generator(device=cuda).manual_seed(rows + c)).to(dtype)
    got = ops.column_sums(a)
    want = a.double().sum(0)
    scale = float(a.double().abs().sum(0).max())
    assert got.dtype == torch.float32 and got.shape == (c,)
    assert float((got.double() - want).abs().max()) <= 2e-6 * scale
    assert torch.equal(got, ops.column_sums(a))                        # fixed order: run-to-run identical


@pytest.mark.parametrize("autocast", [True, False])
def test_odd_channel_conv_bias_gradient(cuda, autocast):
    """A BevConv2d with 59 output channels (DepthNet's depth logits): same output and the same three gradients as
    nn.Conv2d — its bias gradient comes from the column-sum kernel."""
    from omnihd_amd.mm.bricks import BevConv2d
    torch.manual_seed(3)
    ref = torch.nn.Conv2d(64, 59, 1, bias=True).to(cuda).to(memory_format=torch.channels_last)
    m = BevConv2d(64, 59, 1, bias=True).to(cuda).to(memory_format=torch.channels_last)
    m.load_state_dict(ref.state_dict())
    x = torch.randn(3, 64, 10, 14, device=cuda).contiguous(memory_format=torch.channels_last)
    g = torch.randn(3, 59, 10, 14, device=cuda)
    outs = []
    for mod in (m, ref):
        xi = x.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            y = mod(xi)
        grads = torch.autograd.grad(y, [xi, mod.weight, mod.bias], g.to(y.dtype))
        outs.append((y, *grads))
    assert "ConvBiasColsum" in type(outs[0][0].grad_fn).__name__
    tol = 2e-2 if autocast else 1e-5
    for a, b in zip(*outs):
        assert a.dtype == b.dtype and a.shape == b.shape
        assert float((a.float() - b.float()).abs().max()) <= tol * float(b.float().abs().max())


@pytest.mark.parametrize("autocast", [True, False])
def test_odd_channel_conv_result_may_be_modified_in_place(cuda, autocast):
    """the result of a biased BevConv2d with an odd width is a fresh tensor, not a view handed out by a custom
    Function — a ReLU(inplace=True) behind it (mmcv's ConvModule default) works and gives nn.Conv2d's gradients."""
    from omnihd_amd.mm.bricks import use_bev_conv
    torch.manual_seed(4)
    mk = lambda: torch.nn.Sequential(torch.nn.Conv2d(64, 59, 3, padding=1, bias=True), torch.nn.ReLU(inplace=True)).to(cuda).to(
        memory_format=torch.channels_last)
    ref, m = mk(), mk()
    m.load_state_dict(ref.state_dict())
    assert use_bev_conv(m) == 1
    x = torch.randn(2, 64, 9, 11, device=cuda).contiguous(memory_format=torch.channels_last)
    res = []
    for mod in (m, ref):
        xi = x.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            y = mod(xi)
            z = y.float().sigmoid().clamp(0.1, 0.9)          # (the in-place consumer is the ReLU(inplace=True) behind the conv)
        z.sum().backward()
        res.append((y.detach().float(), xi.grad, mod[0].weight.grad, mod[0].bias.grad))
    tol = 2e-2 if autocast else 1e-5
    for a, b in zip(*res):
        assert float((a.float() - b.float()).abs().max()) <= tol * float(b.float().abs().max())

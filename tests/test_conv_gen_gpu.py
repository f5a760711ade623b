"""GPU parity of the GENERAL implicit-GEMM convolution kernel (csrc/conv_gen.hip): strided / padded / dilated forward and data
gradient, transposed convolutions with kernel == stride, channel counts that are multiples of 8 only — in the bf16 form against
torch's fp32 convolution on the same bf16-rounded operands (one bf16 rounding of the result), and in the fp32-grade split form
against torch's fp32 convolution on the fp32 operands (1e-4).  The layers it serves: SECOND's stride-2 stage entries
(reference config projects/configs/bevfusion_NewScenes/bevfusion.py:62-68), ResNet-50's strided 3x3 / 1x1 layers
(bevfusion.py:77-85), SECONDFPN's transposed convolutions (bevfusion.py:69-74), the anchor head's 16/32/72-channel layers."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


GEOMS = [  # B, H, W, cin, cout, k, stride, pad, dil
    (1, 320, 480, 64, 64, 3, 2, 1, 1),        # SECOND stage 0 entry
    (1, 160, 240, 64, 128, 3, 2, 1, 1),       # SECOND stage 1 entry
    (1, 80, 120, 128, 256, 3, 2, 1, 1),       # SECOND stage 2 entry
    (6, 64, 176, 128, 128, 3, 2, 1, 1),       # ResNet-50 layer2.0.conv2
    (6, 64, 176, 256, 512, 1, 2, 0, 1),       # ResNet-50 layer2.0.downsample (three of four input classes get zeros)
    (2, 33, 51, 72, 40, 3, 2, 1, 1),          # odd sizes, channel counts that are multiples of 8 only
    (2, 17, 23, 32, 16, 3, 1, 1, 1),          # stride 1 on 32 channels (masked chunks of a 64-channel K-step)
    (1, 30, 41, 64, 24, 3, 1, 2, 2),          # dilation 2
    (1, 21, 19, 40, 64, 3, 3, 0, 1),          # stride 3, no padding: nine input classes
    (2, 16, 20, 96, 48, 2, 2, 0, 1),          # kernel == stride 2 (the convolution behind SECONDFPN's x2 block)
    (1, 16, 24, 128, 256, 4, 4, 0, 1),        # kernel == stride 4 (x4 block): 16 taps / 16 classes
    (1, 9, 7, 8, 8, 1, 1, 0, 1),              # tiny
    (1, 40, 60, 384, 72, 1, 1, 0, 1),         # anchor head regression branch: data gradient reads 72 channels
]


def _operands(cuda, B, H, W, cin, cout, k):
    torch.manual_seed(B * 1000 + H * 7 + cin + cout + k)
    x = torch.randn(B, cin, H, W, device=cuda).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device=cuda) * (2.0 / (cin * k * k)) ** 0.5).contiguous(memory_format=torch.channels_last)
    return x, w


@pytest.mark.parametrize("B,H,W,cin,cout,k,s,p,d", GEOMS)
def test_split_form_matches_fp32_convolution(cuda, B, H, W, cin, cout, k, s, p, d):
    from omnihd_amd import ops
    x, w = _operands(cuda, B, H, W, cin, cout, k)
    bias = torch.randn(cout, device=cuda)
    want = F.conv2d(x, w, bias, stride=s, padding=p, dilation=d)
    assert ops.conv_gen_supported(0, x.shape, cout, k, s, p, d) and ops.conv_gen_supported(1, x.shape, cout, k, s, p, d)
    got = ops.conv_gen(0, ops.split_f32(x), ops.split_f32(w), bias, tuple(x.shape), cout, k, s, p, d)
    assert got.dtype == torch.float32 and got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert _rel(got, want) <= 1e-4, _rel(got, want)
    assert torch.equal(got, ops.conv_gen(0, ops.split_f32(x), ops.split_f32(w), bias, tuple(x.shape), cout, k, s, p, d))   # run-to-run identical
    g = torch.randn_like(want)
    want_gx = torch.nn.grad.conv2d_input(x.shape, w, g, stride=s, padding=p, dilation=d)
    wt = ops.split_dgrad_weights(ops.split_f32(w))
    got_gx = ops.conv_gen(1, ops.split_f32(g), wt, None, tuple(x.shape), cout, k, s, p, d)
    assert got_gx.shape == x.shape and got_gx.dtype == torch.float32
    assert _rel(got_gx, want_gx) <= 1e-4, _rel(got_gx, want_gx)
    assert torch.equal(want_gx == 0, got_gx == 0) or _rel(got_gx, want_gx) <= 1e-5      # pixels no tap reaches are exact zeros
    assert torch.equal(got_gx, ops.conv_gen(1, ops.split_f32(g), wt, None, tuple(x.shape), cout, k, s, p, d))


@pytest.mark.parametrize("B,H,W,cin,cout,k,s,p,d", GEOMS)
def test_bf16_form_matches_fp32_convolution_of_the_rounded_operands(cuda, B, H, W, cin, cout, k, s, p, d):
    from omnihd_amd import ops
    x, w = _operands(cuda, B, H, W, cin, cout, k)
    xb, wb = x.to(torch.bfloat16), w.to(torch.bfloat16)
    bias = torch.randn(cout, device=cuda)
    want = F.conv2d(xb.float(), wb.float(), bias, stride=s, padding=p, dilation=d)
    got = ops.conv_gen(0, xb, wb, bias, tuple(x.shape), cout, k, s, p, d)
    assert got.dtype == torch.bfloat16 and got.shape == want.shape
    assert _rel(got.float(), want) <= 2.0 ** -7, _rel(got.float(), want)                  # one bf16 rounding of the result
    gb = torch.randn_like(want).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    want_gx = torch.nn.grad.conv2d_input(x.shape, wb.float(), gb.float(), stride=s, padding=p, dilation=d)
    got_gx = ops.conv_gen(1, gb, ops.conv_dgrad_weights(wb), None, tuple(x.shape), cout, k, s, p, d)
    assert _rel(got_gx.float(), want_gx) <= 2.0 ** -7, _rel(got_gx.float(), want_gx)


@pytest.mark.parametrize("B,H,W,cin_t,cout_t,k", [(1, 80, 120, 128, 128, 2), (1, 40, 60, 256, 128, 4), (2, 13, 9, 64, 32, 2), (1, 20, 30, 64, 128, 1)])
def test_transposed_convolution_with_kernel_equal_stride(cuda, B, H, W, cin_t, cout_t, k, monkeypatch):
    """SECONDFPN's up-sampling blocks through the module path of the fp32 step (bricks.BevConvTranspose2d -> ops.deconv_split):
    forward, input gradient and weight gradient against torch's conv_transpose2d in fp32."""
    from omnihd_amd import ops
    from omnihd_amd.mm.bricks import BevConvTranspose2d
    monkeypatch.setenv("OMNIHD_FP32_CONV", "split")
    torch.manual_seed(k * 100 + cin_t)
    m = BevConvTranspose2d(cin_t, cout_t, k, stride=k, bias=False).to(cuda)
    x = torch.randn(B, cin_t, H, W, device=cuda).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = m(x)
    want = F.conv_transpose2d(x.detach(), m.weight.detach(), stride=k)
    assert y.shape == want.shape and _rel(y, want) <= 1e-4, _rel(y, want)
    assert y.grad_fn is not None and "DeconvSplit" in type(y.grad_fn).__name__
    g = torch.randn_like(want)
    y.backward(g)
    xr = x.detach().clone().requires_grad_()
    wr = m.weight.detach().clone().requires_grad_()
    F.conv_transpose2d(xr, wr, stride=k).backward(g)
    assert _rel(x.grad, xr.grad) <= 1e-4 and _rel(m.weight.grad, wr.grad) <= 1e-4, (_rel(x.grad, xr.grad), _rel(m.weight.grad, wr.grad))
    # bf16 form of the same layer (the autocast step): forward and input gradient on the general kernel
    monkeypatch.setenv("OMNIHD_CONV_POLICY", "hip")
    xb = x.detach().to(torch.bfloat16).requires_grad_()
    yb = ops.deconv_hip_wgrad(xb, m.weight, k)
    wantb = F.conv_transpose2d(xb.detach().float(), m.weight.detach().to(torch.bfloat16).float(), stride=k)
    assert _rel(yb.float(), wantb) <= 2.0 ** -7
    yb.backward(g.to(torch.bfloat16))
    assert _rel(xb.grad.float(), xr.grad) <= 3e-2


def test_strided_layers_run_on_the_general_kernel_in_the_fp32_module_path(cuda, monkeypatch):
    """bricks.BevConv2d with stride 2 under the fp32 policy 'split': forward, input gradient and weight gradient come from this
    library's kernels (no library convolution in the autograd graph) and match torch's fp32 convolution to 1e-4."""
    from omnihd_amd.mm.bricks import use_bev_conv
    monkeypatch.setenv("OMNIHD_FP32_CONV", "split")
    torch.manual_seed(9)
    conv = torch.nn.Conv2d(64, 128, 3, stride=2, padding=1, bias=False).to(cuda).to(memory_format=torch.channels_last)
    assert use_bev_conv(conv) == 1
    x = torch.randn(1, 64, 160, 240, device=cuda).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = conv(x)
    assert "ConvSplit" in type(y.grad_fn).__name__
    g = torch.randn_like(y)
    y.backward(g)
    xr = x.detach().clone().requires_grad_()
    wr = conv.weight.detach().clone().requires_grad_()
    yr = F.conv2d(xr, wr, stride=2, padding=1)
    yr.backward(g)
    assert _rel(y, yr) <= 1e-4 and _rel(x.grad, xr.grad) <= 1e-4 and _rel(conv.weight.grad, wr.grad) <= 1e-4


def test_unsupported_geometries_are_refused(cuda):
    from omnihd_amd import ops
    assert not ops.conv_gen_supported(0, (1, 12, 16, 16), 8, 3, 1, 1, 1)          # 12 source channels: not a whole 16-byte chunk
    assert not ops.conv_gen_supported(1, (1, 16, 16, 16), 12, 3, 1, 1, 1)         # data gradient reads cout = 12 channels
    assert not ops.conv_gen_supported(0, (1, 16, 16, 16), 8, 5, 1, 2, 1)          # 25 taps
    assert not ops.conv_gen_supported(0, (1, 16, 16, 16), 8, 3, 5, 1, 1)          # 25 stride classes
    x = torch.randn(1, 16, 8, 8, device=cuda).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(8, 16, 3, 3, device=cuda).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with pytest.raises(ValueError):
        ops.conv_gen(0, x, w, None, (1, 16, 9, 8), 8, 3, 1, 1, 1)                  # source does not have the stated shape

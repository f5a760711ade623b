"""GPU parity of the fp32-grade convolutions on the bf16 matrix cores (csrc/conv_igemm.hip, SPLIT): every fp32 operand as
two bf16 planes, three MFMA products per term, fp32 accumulation — against torch's fp32 convolution on the SAME fp32 operands
(not bf16-rounded ones).  Bounds: forward / data gradient / weight gradient <= 1e-4 of the largest
reference value; the reference layers: BEV encoder cam_stream_lss_bevpoolv2_depthnet.py:201-214, fusion conv
bevf_faster_rcnn_bevdepth.py:61-72, fp32 recipe bevfusion.py:223-268."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def test_split_planes_reconstruct_the_value(cuda):
    from omnihd_amd import ops
    torch.manual_seed(0)
    x = torch.randn(3, 7, 5, 13, device=cuda) * torch.logspace(-20, 20, 13, device=cuda)
    x.view(-1)[5] = float("inf"); x.view(-1)[9] = float("nan"); x.view(-1)[11] = 0.0; x.view(-1)[12] = -0.0
    hi, lo = ops.split_f32(x)
    assert hi.dtype == lo.dtype == torch.bfloat16 and hi.shape == x.shape and hi.stride() == x.stride()
    rec = hi.float() + lo.float()
    ok = torch.isfinite(x)
    assert float(((rec - x)[ok].abs() / x[ok].abs().clamp_min(1e-30)).max()) <= 2.0 ** -16
    assert torch.isinf(hi.view(-1)[5]) and lo.view(-1)[5] == 0 and torch.isnan(hi.view(-1)[9]) and lo.view(-1)[9] == 0
    assert torch.equal(hi[ok], x.to(torch.bfloat16)[ok])           # hi is torch's own bf16 rounding
    xc = torch.randn(2, 64, 6, 10, device=cuda).contiguous(memory_format=torch.channels_last)
    h2, l2 = ops.split_f32(xc)
    assert h2.is_contiguous(memory_format=torch.channels_last) and torch.equal(h2, xc.to(torch.bfloat16))


GEOMS = [  # B, H, W, cin, cout, k, dil, tile
    (1, 160, 240, 128, 256, 3, 1, 300), (1, 160, 240, 128, 256, 3, 1, 256), (1, 160, 240, 128, 256, 3, 1, 254),
    (1, 160, 240, 128, 256, 3, 1, 128), (1, 160, 240, 64, 64, 3, 1, 0), (2, 33, 50, 192, 136, 3, 2, 300),
    (6, 64, 176, 256, 256, 3, 6, 300), (1, 37, 41, 128, 72, 3, 12, 0), (2, 20, 30, 256, 128, 1, 1, 0),
    (1, 64, 176, 1280, 256, 1, 1, 0), (1, 9, 7, 64, 8, 3, 1, 0),
]


@pytest.mark.parametrize("B,H,W,cin,cout,k,dil,tile", GEOMS)
def test_split_forward_and_data_gradient_match_fp32_convolution(cuda, B, H, W, cin, cout, k, dil, tile):
    from omnihd_amd import ops
    torch.manual_seed(B * H + cin + k)
    x = torch.randn(B, cin, H, W, device=cuda).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device=cuda) * (2.0 / (cin * k * k)) ** 0.5).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(cout, device=cuda)
    pad = dil * (k // 2)
    want = F.conv2d(x, w, bias, padding=pad, dilation=dil)
    got = ops.conv_fwd_split(ops.split_f32(x), ops.split_f32(w), bias, dil, tile)
    assert got.dtype == torch.float32 and got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert _rel(got, want) <= 1e-4, _rel(got, want)
    assert torch.equal(got, ops.conv_fwd_split(ops.split_f32(x), ops.split_f32(w), bias, dil, tile))     # deterministic
    if cout % 64 == 0:
        g = torch.randn_like(want)
        want_gx = torch.nn.grad.conv2d_input(x.shape, w, g, padding=pad, dilation=dil)
        wt = ops.split_dgrad_weights(ops.split_f32(w))
        got_gx = ops.conv_fwd_split(ops.split_f32(g), wt, None, dil, tile)
        assert _rel(got_gx, want_gx) <= 1e-4, _rel(got_gx, want_gx)


def test_split_is_far_closer_to_fp32_than_one_bf16_product(cuda):
    """What the third of the bf16 rate buys: the same layer through the plain bf16 kernel is ~100x further from fp32."""
    from omnihd_amd import ops
    torch.manual_seed(1)
    x = torch.randn(1, 256, 40, 60, device=cuda).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(128, 256, 3, 3, device=cuda) * 0.03).contiguous(memory_format=torch.channels_last)
    want = F.conv2d(x, w, None, padding=1)
    split = ops.conv_fwd_split(ops.split_f32(x), ops.split_f32(w))
    plain = ops.conv_fwd(x.to(torch.bfloat16), w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)).float()
    assert _rel(split, want) <= 3e-5 and _rel(plain, want) >= 30 * _rel(split, want)


@pytest.mark.parametrize("policy", ["split", "tune"])
def test_fp32_module_path_gradients_match_nn_conv2d(cuda, policy, monkeypatch):
    """BevConv2d on fp32 activations without autocast (the reference-precision step): output and all three gradients against
    nn.Conv2d in fp32, 1e-4; a strided layer (forward / data gradient stay on MIOpen, weight gradient on the split chain),
    a biased one and a 1x1."""
    from omnihd_amd.mm.bricks import use_bev_conv
    monkeypatch.setenv("OMNIHD_FP32_CONV", policy)
    torch.manual_seed(2)
    # (smooth activations: behind a ReLU an output within 1e-5 of zero flips its mask between two fp32-grade implementations
    # and moves single gradient entries by their full size — a property of the comparison, not of either kernel)
    mk = lambda: torch.nn.Sequential(torch.nn.Conv2d(64, 128, 3, padding=1, bias=False), torch.nn.Tanh(),
                                     torch.nn.Conv2d(128, 128, 3, stride=2, padding=1, bias=True), torch.nn.Tanh(),
                                     torch.nn.Conv2d(128, 64, 1, bias=True), torch.nn.Tanh(),
                                     torch.nn.Conv2d(64, 64, 3, padding=2, dilation=2, bias=False)).to(cuda).to(
                                         memory_format=torch.channels_last)
    ref, m = mk(), mk()
    m.load_state_dict(ref.state_dict())
    assert use_bev_conv(m) == 4
    x = torch.randn(2, 64, 24, 40, device=cuda).contiguous(memory_format=torch.channels_last)
    res = []
    for mod in (m, ref):
        mod.train()
        xi = x.clone().requires_grad_()
        y = mod(xi)
        y.square().mean().backward()
        res.append([y.detach(), xi.grad] + [p.grad for p in mod.parameters()])
    if policy == "split":
        assert "ConvSplit" in type(m(x).grad_fn).__name__
    for a, b in zip(*res):
        assert a.dtype == torch.float32 and _rel(a, b) <= 1e-4, _rel(a, b)


def test_split_weight_gradient_of_a_bev_sized_layer(cuda, monkeypatch):
    from omnihd_amd import ops
    monkeypatch.setenv("OMNIHD_FP32_CONV", "split")
    torch.manual_seed(3)
    x = torch.randn(1, 128, 160, 240, device=cuda).contiguous(memory_format=torch.channels_last).requires_grad_()
    w = (torch.randn(256, 128, 3, 3, device=cuda) * 0.03).requires_grad_()
    g = torch.randn(1, 256, 160, 240, device=cuda) * 0.01
    y = ops.conv_split(x, w, None, (1, 1), (1, 1))
    gx, gw = torch.autograd.grad(y, [x, w], g)
    x2, w2 = x.detach().clone().requires_grad_(), w.detach().clone().requires_grad_()
    wx, ww = torch.autograd.grad(F.conv2d(x2, w2, None, padding=1), [x2, w2], g)
    assert _rel(gx, wx) <= 1e-4 and _rel(gw, ww) <= 1e-4, (_rel(gx, wx), _rel(gw, ww))


@pytest.mark.parametrize("B,H,W,cin,cout", [
    (2, 48, 47, 512, 256),      # W + 1 a multiple of 8: exactly one zero column per padded image row; two images (row offsets cross them)
    (1, 40, 60, 520, 264),      # channel counts that are not multiples of the 128-wide tiles
    (3, 33, 50, 384, 384),      # odd image height, three images
    (1, 160, 240, 640, 384),    # the fusion convolution of the detector
])
def test_split_weight_gradient_with_register_shifted_taps(cuda, B, H, W, cin, cout):
    """3x3 / dilation 1 weight gradient on split operands, every tap against the fp32 library gradient, 1e-4: the three-taps form of
    csrc/conv_wgrad_nhwc.hip (a 34-row X tile read at three pixel shifts, padded raster)."""
    from omnihd_amd import ops
    torch.manual_seed(B * 1000 + W)
    x = torch.randn(B, cin, H, W, device=cuda).contiguous(memory_format=torch.channels_last)
    g = (torch.randn(B, cout, H, W, device=cuda) * 0.05).contiguous(memory_format=torch.channels_last)
    dw = ops.conv_wgrad_split(ops.split_f32(x), ops.split_f32(g), 3, 1, 1, 1)
    w0 = torch.zeros(cout, cin, 3, 3, device=cuda)
    ref = torch.ops.aten.convolution_backward(g, x, w0, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    dw = dw.reshape(ref.shape) if dw.shape != ref.shape else dw
    for ky in range(3):
        for kx in range(3):
            assert _rel(dw[:, :, ky, kx], ref[:, :, ky, kx]) <= 1e-4, (ky, kx, _rel(dw[:, :, ky, kx], ref[:, :, ky, kx]))


def test_batched_weight_images_equal_the_layer_by_layer_path(cuda):
    """ops.refresh_split_shadows(): ONE launch refreshes the forward and data-gradient planes of every registered layer; bit for bit
    what split_weight builds layer by layer (layout copy + split + two transposes), for OIHW and channels_last parameters."""
    from omnihd_amd import ops
    torch.manual_seed(11)
    ws = [torch.nn.Parameter(torch.randn(s, device=cuda) * 0.1) for s in [(64, 128, 3, 3), (40, 64, 1, 1), (256, 192, 3, 3), (8, 64, 3, 3)]]
    ws.append(torch.nn.Parameter((torch.randn(96, 64, 3, 3, device=cuda) * 0.1).contiguous(memory_format=torch.channels_last)))
    for i, w in enumerate(ws):
        ops.split_weight(w)
        if i != 1:
            ops.split_weight(w, dgrad=True)           # layer 1 never asked for its data-gradient image
    with torch.no_grad():
        for w in ws:
            w.mul_(1.5).add_(0.01)                    # the optimiser step: versions move on
    assert ops.refresh_split_shadows() == len(ws)
    got = [(ops.split_weight(w), ops.split_weight(w, dgrad=True) if i != 1 else None) for i, w in enumerate(ws)]
    got = [tuple(None if p is None else tuple(t.clone() for t in p) for p in g) for g in got]
    assert ops.refresh_split_shadows() == 0           # nothing stale any more
    ops._SPLIT_SHADOW.clear()
    for i, w in enumerate(ws):
        f = ops.split_weight(w)
        assert all(torch.equal(a, b) for a, b in zip(got[i][0], f)), i
        if i != 1:
            d = ops.split_weight(w, dgrad=True)
            assert all(torch.equal(a, b) for a, b in zip(got[i][1], d)), i


def test_batchnorm_hands_its_planes_to_the_next_split_convolution(cuda, monkeypatch):
    """conv -> BatchNorm(train)+ReLU -> conv in fp32: from the second step on the fused BatchNorm kernels write the bf16 planes of
    their output (forward) and of their input gradient (backward) and the convolutions skip their own split pass; results are
    bit-identical to the run without the hand-over."""
    from omnihd_amd import ops
    from omnihd_amd.mm.bricks import run_fused, use_bev_conv
    monkeypatch.setenv("OMNIHD_FP32_CONV", "split")

    def run(handover, steps=3):
        monkeypatch.setenv("OMNIHD_SPLIT_HANDOVER", "1" if handover else "0")
        ops._PLANES_WANTED.clear(); ops._PLANES_UNUSED.clear()
        for k in ops.HANDOVER_STATS:
            ops.HANDOVER_STATS[k] = 0
        torch.manual_seed(4)
        net = torch.nn.Sequential(torch.nn.Conv2d(64, 128, 3, padding=1, bias=False), torch.nn.BatchNorm2d(128), torch.nn.ReLU(),
                                  torch.nn.Conv2d(128, 64, 3, padding=1, bias=False), torch.nn.BatchNorm2d(64), torch.nn.ReLU(),
                                  torch.nn.Conv2d(64, 64, 1, bias=False)).to(cuda).to(memory_format=torch.channels_last)
        use_bev_conv(net)
        x = torch.randn(2, 64, 24, 40, device=cuda).contiguous(memory_format=torch.channels_last)
        outs = []
        for _ in range(steps):
            net.zero_grad(set_to_none=True)
            xi = x.clone().requires_grad_()
            y = run_fused(net, xi)
            y.square().mean().backward()
            outs.append([y.detach().clone(), xi.grad.clone()] + [p.grad.clone() for p in net.parameters()])
        return outs, dict(ops.HANDOVER_STATS)

    base, _ = run(False)
    got, stats = run(True)
    for a, b in zip(base, got):
        for u, v in zip(a, b):
            assert torch.equal(u, v)
    assert stats["taken"] >= 4, stats           # steps 2 and 3: at least the two forward hand-overs each


def test_frozen_batchnorm_epilogue_hands_its_planes_to_the_next_split_convolution(cuda, monkeypatch):
    """The same hand-over from the frozen-BatchNorm epilogue (affine + ReLU: every BatchNorm of the image backbone): from the second
    step on it writes the hi / lo planes of its output and the convolution behind it skips its split pass; bit-identical."""
    from omnihd_amd import ops
    monkeypatch.setenv("OMNIHD_FP32_CONV", "split")
    torch.manual_seed(9)
    w1 = (torch.randn(128, 64, 3, 3, device=cuda) * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_()
    w2 = (torch.randn(64, 128, 3, 3, device=cuda) * 0.03).contiguous(memory_format=torch.channels_last).requires_grad_()
    scale, shift = torch.rand(128, device=cuda) + 0.5, torch.randn(128, device=cuda) * 0.1
    x0 = torch.randn(2, 64, 24, 40, device=cuda).contiguous(memory_format=torch.channels_last)

    def run(handover, steps=3):
        monkeypatch.setenv("OMNIHD_SPLIT_HANDOVER", "all" if handover else "1")      # (opt-in for this epilogue: see docs/SWITCHES.md)
        ops._PLANES_WANTED.clear(); ops._PLANES_UNUSED.clear()
        for k in ops.HANDOVER_STATS:
            ops.HANDOVER_STATS[k] = 0
        outs = []
        for _ in range(steps):
            x = x0.clone().requires_grad_()
            y = ops.conv_split(x, w1, None, (1, 1), (1, 1))
            y = ops.affine_act(y, scale, shift, None, True)
            y = ops.conv_split(y, w2, None, (1, 1), (1, 1))
            outs.append([y.detach()] + [g.detach() for g in torch.autograd.grad(y.square().mean(), [x, w1, w2])])
        return outs, dict(ops.HANDOVER_STATS)

    base, stats0 = run(False)
    got, stats = run(True)
    assert stats0["taken"] == 0 and stats["taken"] >= 2, (stats0, stats)      # steps 2 and 3
    for a, b in zip(base, got):
        for u, v in zip(a, b):
            assert torch.equal(u, v)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_weight_images_follow_the_fused_optimiser_in_the_step_loop(cuda, dtype):
    """Regression (round 3): torch.optim.AdamW(fused=True) does not move the parameters' version counters, so version-keyed weight
    images went stale after the first step.  In the step loop of the harness every convolution weight image (bf16 shadow / split
    planes / data-gradient image) must equal an image rebuilt from the CURRENT master weight after each optimiser step."""
    from omnihd_amd import ops
    from omnihd_amd.harness import FusionTrainStep
    st = FusionTrainStep(res="tiny", batch=1, radar_dims=7, dtype=dtype)
    for _ in range(3):
        st.step()
    torch.cuda.synchronize()
    checked = 0
    if dtype == "bf16":
        for k, (ref, ver, shadow) in list(ops._BF16_SHADOW.items()):
            w = ref()
            if w is None or not w.requires_grad:
                continue
            assert ver == ops._wver(w)
            assert torch.equal(shadow, w.detach().to(torch.bfloat16)), "stale bf16 image"
            d = ops._BF16_DGRAD.get(k)
            if d is not None and d[0]() is w:
                want = ops.conv_dgrad_weights(w.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
                assert torch.equal(d[2], want), "stale data-gradient image"
            checked += 1
    else:
        for (wid, dgrad), (ref, ver, planes) in list(ops._SPLIT_SHADOW.items()):
            w = ref()
            if w is None or not w.requires_grad:
                continue
            assert ver == ops._wver(w)
            fresh = ops.split_f32(w.detach().float().contiguous(memory_format=torch.channels_last))
            if dgrad:
                fresh = ops.split_dgrad_weights(fresh)
            assert all(torch.equal(a, b) for a, b in zip(planes, fresh)), "stale split planes"
            checked += 1
    assert checked > 10


@pytest.mark.parametrize("weights_channels_last", [False, True])
def test_weight_gradients_on_the_side_stream_equal_the_in_line_ones(cuda, weights_channels_last, monkeypatch):
    """OMNIHD_WGRAD_OVERLAP (one rank; by default armed by the pooling backward for the layers behind it, "all" = every layer,
    what this test uses): the weight gradient of a split convolution is computed on a side stream while the
    data-gradient chain goes on, and the autograd engine's end-of-backward callback joins that stream.  A chain of
    convolutions big enough that the side stream is still busy when ``backward`` returns control to the engine: every gradient
    must equal the in-line run's (same kernels, same order of arithmetic: 1e-6 where MIOpen's atomics decide the last bit),
    for parameters in either memory format (autograd keeps a gradient in the parameter's layout as it is and would COPY any
    other one on the main stream); a parameter that already holds a gradient accumulates correctly (in-line fallback)."""
    from omnihd_amd import ops
    torch.manual_seed(3)
    chans = [64, 128, 256, 256, 128]
    ws = [torch.randn(chans[i + 1], chans[i], 3, 3, device=cuda) * 0.05 for i in range(4)]
    if weights_channels_last:
        ws = [w.contiguous(memory_format=torch.channels_last) for w in ws]
    x0 = torch.randn(2, 64, 96, 160, device=cuda).contiguous(memory_format=torch.channels_last)

    class Arm(torch.autograd.Function):                   # stands in for the pooling backward: the layers in front of it (in
        @staticmethod                                      # backward order) stay in line, what follows goes to the side stream
        def forward(ctx, t):
            return t.view_as(t)

        @staticmethod
        def backward(ctx, gt):
            ops.wgrad_overlap_arm()
            return gt

    def run(overlap, accumulate=False):
        monkeypatch.setenv("OMNIHD_WGRAD_OVERLAP", overlap)
        params = [w.clone().requires_grad_() for w in ws]
        if weights_channels_last:
            assert all(p.is_contiguous(memory_format=torch.channels_last) for p in params)
        x = x0.clone().requires_grad_()
        for rep in range(2 if accumulate else 1):
            y = x
            for i, p in enumerate(params):
                y = torch.relu(ops.conv_split(y, p, None, (1, 1), (1, 1)))
                if i == 1 and overlap == "1":
                    y = Arm.apply(y)
            (y * y).mean().backward()                         # the engine's callback joins the side stream here
        return [p.grad.clone() for p in params] + [x.grad.clone()]

    base = run("0")
    from omnihd_amd import ops as _ops
    for got, want in zip(run("all"), base):
        assert got.shape == want.shape and torch.isfinite(got).all()
        assert _rel(got, want) <= 1e-6
    for got, want in zip(run("1"), base):                     # layers 3, 2 in line, layers 1, 0 (behind the marker) on the side stream
        assert torch.isfinite(got).all() and _rel(got, want) <= 1e-6
    assert not _ops._WGRAD_SIDE_USED and not _ops._WGRAD_ARMED
    base2 = run("0", accumulate=True)
    assert _ops._WGRAD_SIDE and not _ops._WGRAD_SIDE_USED and not _ops._WGRAD_ARMED     # the side stream was used and joined
    for got, want in zip(run("all", accumulate=True), base2):
        assert _rel(got, want) <= 1e-6
    assert _rel(base2[0], 2 * base[0]) <= 1e-5                # two identical passes accumulated


def test_shared_weight_of_two_convolutions_with_the_side_stream(cuda, monkeypatch):
    """a weight used by TWO split convolutions of one graph.  `weight.grad` is None at both backward calls, and
    the engine sums the two gradients on the caller's stream when the second arrives — the first must not still be in flight on
    the side stream then.  From the second sighting of a weight inside a pass the layer stays in line and the caller's stream
    waits for the side stream first; a long kernel parked on the side stream makes a missing wait visible."""
    from omnihd_amd import ops
    torch.manual_seed(11)
    w0 = torch.randn(128, 128, 3, 3, device=cuda) * 0.05
    x0 = torch.randn(2, 128, 96, 160, device=cuda).contiguous(memory_format=torch.channels_last)

    def run(overlap):
        monkeypatch.setenv("OMNIHD_WGRAD_OVERLAP", overlap)
        p = w0.clone().requires_grad_()
        x = x0.clone().requires_grad_()
        y = torch.relu(ops.conv_split(x, p, None, (1, 1), (1, 1)))
        y = torch.relu(ops.conv_split(y, p, None, (1, 1), (1, 1)))            # the same parameter again
        if overlap != "0" and cuda.index in ops._WGRAD_SIDE:
            with torch.cuda.stream(ops._WGRAD_SIDE[cuda.index]):
                torch.cuda._sleep(20_000_000)                                  # ~10 ms in front of whatever goes there next
        (y * y).mean().backward()
        return p.grad.clone(), x.grad.clone()

    base = run("0")
    run("all")                                                                 # creates the side stream
    for _ in range(2):
        got = run("all")
        for g, wnt in zip(got, base):
            assert torch.isfinite(g).all() and _rel(g, wnt) <= 1e-6
    assert not ops._WGRAD_SIDE_USED and not ops._WGRAD_SEEN


@pytest.mark.parametrize("bucket_mb", [1, 25, 100])
def test_side_stream_weight_gradients_under_ddp_go_straight_into_the_bucket_views(cuda, bucket_mb):
    """the N > 1 step must be the N = 1 step.  A chain of split convolutions under DistributedDataParallel
    (one-rank RCCL group, several buckets) with `ops.ddp_wgrad_overlap`: once the reducer has re-bucketed, the weight gradients
    are computed on the side stream INTO the reducer's bucket views, autograd keeps an alias as `.grad` (no copy on the caller's
    stream), and every bucket's all-reduce waits for the side stream.  A long kernel parked on the side stream before each
    backward makes any consumer that does not wait read the PREVIOUS iteration's reduced gradient (different input each
    iteration).  Gradients must equal the plain in-line run's, iteration by iteration.  Child process: the process group must
    not leak into the other tests."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import copy, os, sys, torch, torch.distributed as dist
from torch import nn
sys.path[:0] = [%r, %r]
os.environ["OMNIHD_FP32_CONV"] = "split"
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
from omnihd_amd import ops
from omnihd_amd.mm.bricks import use_bev_conv
torch.manual_seed(5)
chans = [64, 128, 256, 256, 128]
net = nn.Sequential(*[m for i in range(4) for m in (nn.Conv2d(chans[i], chans[i + 1], 3, padding=1, bias=False), nn.ReLU())]).to(dev)
net = net.to(memory_format=torch.channels_last)
use_bev_conv(net)
ref = copy.deepcopy(net)
xs = [torch.randn(2, 64, 96, 160, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(5)]

def grads_of(model, x):
    for p in model.parameters():
        p.grad = None
    y = model(x)
    (y * y).mean().backward()
    torch.cuda.synchronize()
    return [p.grad.clone() for p in model.parameters()]

os.environ["OMNIHD_WGRAD_OVERLAP"] = "0"
want = [grads_of(ref, x) for x in xs]
os.environ["OMNIHD_WGRAD_OVERLAP"] = "all"
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
ddp = nn.parallel.DistributedDataParallel(net, device_ids=[0], broadcast_buffers=False, bucket_cap_mb=%d, gradient_as_bucket_view=True)
assert ops.ddp_wgrad_overlap(ddp)
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
worst = 0.0
for it, x in enumerate(xs):
    for p in ddp.parameters():
        p.grad = None
    if 0 in ops._WGRAD_SIDE:
        with torch.cuda.stream(ops._WGRAD_SIDE[0]):
            torch.cuda._sleep(20_000_000)
    y = ddp(x)
    (y * y).mean().backward()
    got = [p.grad for p in net.parameters()]          # read on the caller's stream right behind backward, like clip / AdamW
    got = [g.clone() for g in got]
    torch.cuda.synchronize()
    info = ops.ddp_overlap_info()
    for g, w in zip(got, want[it]):
        assert torch.isfinite(g).all()
        worst = max(worst, rel(g, w))
    if it >= 2:
        for p in net.parameters():
            v = ops._ddp_bucket_view(p)
            assert v is not None and p.grad.data_ptr() == v.data_ptr() and p.grad.stride() == p.stride()
info = ops.ddp_overlap_info()
assert info["hooked"] and info["settled"] and info["views"] == 4 and info["direct_writes"] >= 4 * 2, info
assert worst <= 1e-6, worst
with ddp.no_sync():                                    # no synchronisation this pass: the views must not be used
    assert all(ops._ddp_bucket_view(p) is None for p in net.parameters())
dist.destroy_process_group()
print("DDP_OVERLAP_OK", worst, info)
''' % (root, os.path.join(root, "omnihd-scenes_amd"), bucket_mb)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 90), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "DDP_OVERLAP_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


@pytest.mark.parametrize("overlap", ["all", "1", "0"])
def test_shared_weight_under_ddp_bucket_views_sums_both_uses(cuda, overlap):
    """ONE weight used by TWO convolutions of a backward pass under the hooked reducer.  Both uses used
    to write their gradient into the reducer's bucket view of that weight (``weight.grad`` is still None while the first alias is
    pending), so autograd summed two aliases of one buffer: 2*g2 instead of g1 + g2 — and with the first use in line and the second
    on the side stream also a cross-stream race.  Now the second sighting computes into a fresh tensor.  Gradients must equal the
    plain in-line run's for side stream on all layers / behind the pooling backward only / off."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import copy, os, sys, torch, torch.distributed as dist
from torch import nn
sys.path[:0] = [%r, %r]
os.environ["OMNIHD_FP32_CONV"] = "split"
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
from omnihd_amd import ops
from omnihd_amd.mm.bricks import use_bev_conv
torch.manual_seed(9)

class Twice(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Conv2d(64, 128, 3, padding=1, bias=False)
        self.shared = nn.Conv2d(128, 128, 3, padding=1, bias=False)
        self.b = nn.Conv2d(128, 64, 3, padding=1, bias=False)
    def forward(self, x):
        y = torch.relu(self.a(x))
        y = torch.relu(self.shared(y))
        y = torch.relu(self.shared(y) * 0.5 + y)          # the second use sees another input and another gradient
        return self.b(y)

net = Twice().to(dev).to(memory_format=torch.channels_last)
use_bev_conv(net)
ref = copy.deepcopy(net)
xs = [torch.randn(2, 64, 64, 96, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(5)]

def grads_of(model, x):
    for p in model.parameters():
        p.grad = None
    y = model(x)
    (y * y).mean().backward()
    torch.cuda.synchronize()
    return [p.grad.clone() for p in model.parameters()]

os.environ["OMNIHD_WGRAD_OVERLAP"] = "0"
want = [grads_of(ref, x) for x in xs]
os.environ["OMNIHD_WGRAD_OVERLAP"] = %r
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
ddp = nn.parallel.DistributedDataParallel(net, device_ids=[0], broadcast_buffers=False, bucket_cap_mb=1, gradient_as_bucket_view=True)
assert ops.ddp_wgrad_overlap(ddp)
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
worst = 0.0
for it, x in enumerate(xs):
    for p in ddp.parameters():
        p.grad = None
    if 0 in ops._WGRAD_SIDE:
        with torch.cuda.stream(ops._WGRAD_SIDE[0]):
            torch.cuda._sleep(20_000_000)
    y = ddp(x)
    (y * y).mean().backward()
    got = [p.grad.clone() for p in net.parameters()]
    torch.cuda.synchronize()
    for g, w in zip(got, want[it]):
        assert torch.isfinite(g).all()
        worst = max(worst, rel(g, w))
info = ops.ddp_overlap_info()
assert info["hooked"] and info["settled"] and info["direct_writes"] >= 3 * 2, info
assert worst <= 1e-6, worst
dist.destroy_process_group()
print("DDP_SHARED_OK", worst, info)
''' % (root, os.path.join(root, "omnihd-scenes_amd"), overlap)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29800 + os.getpid() % 90), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "DDP_SHARED_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])

#!/usr/bin/env python3
"""Split (3-term bf16, fp32 accumulate) convolution kernels against MIOpen's fp32 kernels on the geometries of the fp32
training step: forward, data gradient, weight gradient (three launches of the bf16 chain), time per call and effective
TFLOP/s (2*M*N*K of the fp32 convolution)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import bench  # noqa: E402,F401
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from omnihd_amd import ops  # noqa: E402

torch.backends.cudnn.benchmark = True
GEOMS = [(1, 160, 240, 1024, 1024, 3, 1), (1, 160, 240, 1024, 512, 3, 1), (1, 160, 240, 512, 512, 3, 1), (1, 160, 240, 512, 256, 3, 1),
         (1, 160, 240, 640, 384, 3, 1), (6, 64, 176, 1024, 256, 3, 1), (6, 64, 176, 256, 256, 3, 1), (6, 64, 176, 256, 256, 3, 6),
         (6, 64, 176, 256, 256, 3, 12), (6, 64, 176, 1280, 256, 1, 1), (6, 64, 176, 256, 256, 1, 1), (1, 160, 240, 64, 64, 3, 1),
         (1, 80, 120, 128, 128, 3, 1), (1, 40, 60, 256, 256, 3, 1), (6, 32, 88, 128, 128, 3, 1), (6, 16, 44, 256, 256, 3, 1),
         (6, 8, 22, 512, 512, 3, 1), (6, 64, 176, 64, 256, 1, 1), (6, 32, 88, 512, 128, 1, 1), (6, 16, 44, 1024, 256, 1, 1)]


def clock(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tot = {"fs": 0, "fm": 0, "ds": 0, "dm": 0, "ws": 0, "wm": 0}
for B, H, W, cin, cout, k, dil in GEOMS:
    x = torch.randn(B, cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device="cuda") * 0.02).contiguous(memory_format=torch.channels_last)
    pad = dil * (k // 2)
    g = torch.randn(B, cout, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    flops = 2.0 * B * H * W * cin * cout * k * k
    xs, ws, gs = ops.split_f32(x), ops.split_f32(w), ops.split_f32(g)
    wt = ops.split_dgrad_weights(ws)
    t_sp = clock(lambda: ops.split_f32(x))
    f_s = clock(lambda: ops.conv_fwd_split(xs, ws, None, dil))
    f_m = clock(lambda: F.conv2d(x, w, None, padding=pad, dilation=dil))
    err = float((ops.conv_fwd_split(xs, ws, None, dil) - F.conv2d(x, w, None, padding=pad, dilation=dil)).abs().max() /
                F.conv2d(x, w, None, padding=pad, dilation=dil).abs().max())
    d_s = clock(lambda: ops.conv_fwd_split(gs, wt, None, dil)) if cout % 64 == 0 else float("nan")
    d_m = clock(lambda: torch.nn.grad.conv2d_input(x.shape, w, g, padding=pad, dilation=dil))
    w_s = clock(lambda: ops.conv_wgrad_split(xs, gs, k, 1, pad, dil))
    w_m = clock(lambda: torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [pad, pad], [dil, dil], False, [0, 0], 1,
                                                            [False, True, False])[1])
    tf = lambda ms: flops / ms / 1e9
    print(f"{B}x{H}x{W} {cin:4d}->{cout:4d} k{k} d{dil:2d}: split pass {t_sp*1e3:6.1f} us | fwd split {f_s:6.3f} ms {tf(f_s):5.0f} TF  miopen {f_m:6.3f} ms {tf(f_m):4.0f} TF"
          f" | dgrad split {d_s:6.3f} ms {tf(d_s):5.0f} TF  miopen {d_m:6.3f} ms {tf(d_m):4.0f} TF | wgrad split {w_s:6.3f} ms {tf(w_s):5.0f} TF  miopen {w_m:6.3f} ms {tf(w_m):4.0f} TF"
          f" | fwd err {err:.1e}", flush=True)
    for key, v in (("fs", f_s), ("fm", f_m), ("ds", d_s if d_s == d_s else d_m), ("dm", d_m), ("ws", w_s), ("wm", w_m)):
        tot[key] += v
print("sums (ms): fwd split %.2f miopen %.2f | dgrad split %.2f miopen %.2f | wgrad split %.2f miopen %.2f" %
      (tot["fs"], tot["fm"], tot["ds"], tot["dm"], tot["ws"], tot["wm"]))

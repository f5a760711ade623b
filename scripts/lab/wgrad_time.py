#!/usr/bin/env python3
"""Weight-gradient chain (k_to_kmajor + k_wgrad_mfma_glds[3] + k_sum_slabs) on the 3x3 geometries of the step, against MIOpen.
OMNIHD_WGRAD_3TAPS=0 selects the one-tap-per-workgroup kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import bench  # noqa: F401
import torch
from omnihd_amd import ops
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")


def clock(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


geos = [(1, 160, 240, 1024, 1024, 1), (1, 160, 240, 1024, 512, 1), (1, 160, 240, 512, 512, 1), (1, 160, 240, 512, 256, 1),
        (1, 160, 240, 640, 384, 1), (6, 64, 176, 1024, 256, 1), (6, 64, 176, 256, 256, 1), (6, 64, 176, 256, 256, 6), (6, 32, 88, 128, 128, 1),
        (6, 16, 44, 256, 256, 1), (1, 40, 60, 256, 256, 1)]
tot = [0.0, 0.0]
for B, H, W, cin, cout, d in geos:
    x = torch.randn(B, cin, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    g = torch.randn(B, cout, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    flops = 2.0 * B * H * W * cin * cout * 9
    t0 = clock(lambda: ops.conv_wgrad(x, g, 3, 1, d, d))
    t1 = clock(lambda: torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [d, d], [d, d], False, [0, 0], 1, [False, True, False]))
    ref = torch.ops.aten.convolution_backward(g.float(), x.float(), w.float(), None, [1, 1], [d, d], [d, d], False, [0, 0], 1,
                                              [False, True, False])[1]
    got = ops.conv_wgrad(x, g, 3, 1, d, d)
    err = float((got - ref).abs().max() / ref.abs().max())
    tot[0] += t0; tot[1] += t1
    print(f"{B}x{H}x{W} {cin:4d}->{cout:4d} dil {d}: ours {t0*1e6:7.1f} us {flops/t0/1e12:5.0f} TF | miopen {t1*1e6:7.1f} us {flops/t1/1e12:5.0f} TF | "
          f"max err vs fp32 {err:.1e}", flush=True)
print(f"sum: ours {tot[0]*1e3:.2f} ms, miopen {tot[1]*1e3:.2f} ms")

#!/usr/bin/env python3
"""BEV-sized convolutions: implicit-GEMM MFMA kernel (csrc/conv_igemm.hip) vs MIOpen (find mode), forward and data gradient."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from omnihd_amd import ops
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")


def clock(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


geos = [(1, 160, 240, 1024, 1024, 3), (1, 160, 240, 512, 512, 3)]
if len(sys.argv) > 1 and sys.argv[1] == "all":
    geos = [(1, 160, 240, 1024, 1024, 3), (1, 160, 240, 1024, 512, 3), (1, 160, 240, 512, 512, 3), (1, 160, 240, 512, 256, 3),
            (1, 160, 240, 640, 384, 3), (6, 64, 176, 1024, 256, 3), (6, 64, 176, 256, 256, 3), (6, 64, 176, 256, 256, 1)]
for B, H, W, cin, cout, k in geos:
    x = torch.randn(B, cin, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device=dev) * 0.02).bfloat16().contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, cout, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    flops = 2.0 * B * H * W * cin * cout * k * k
    wt = ops.conv_dgrad_weights(w) if cout % 64 == 0 else None
    line = f"{B}x{H}x{W} {cin:4d}->{cout:4d} k{k}:"
    for tile in (256, 254, 300):
        t = clock(lambda: ops.conv_fwd(x, w, None, 1, tile))
        line += f"  fwd{tile} {t*1e3:6.3f} ms {flops/t/1e12:6.0f} TF"
    t = clock(lambda: torch.nn.functional.conv2d(x, w, None, 1, k // 2))
    line += f" | miopen fwd {t*1e3:6.3f} ms {flops/t/1e12:6.0f} TF"
    if wt is not None:
        t = clock(lambda: ops.conv_fwd(gy, wt, None, 1, 0))
        line += f" | dgrad {t*1e3:6.3f} ms {flops/t/1e12:6.0f} TF"
        t = clock(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [k // 2, k // 2], [1, 1], False, [0, 0], 1,
                                                              [True, False, False])[0])
        line += f" | miopen dgrad {t*1e3:6.3f} ms {flops/t/1e12:6.0f} TF"
    print(line, flush=True)

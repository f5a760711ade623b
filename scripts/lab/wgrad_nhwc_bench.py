#!/usr/bin/env python3
"""Weight gradient per geometry: the NHWC kernel (csrc/conv_wgrad_nhwc.hip) vs the staged chain (csrc/conv_wgrad.hip) vs the library's
fp32 kernel, split form, within one process (events around batches of calls; the library in find mode)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
from omnihd_amd.harness import seed_miopen_db
seed_miopen_db()
import torch
from omnihd_amd import ops
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
GEOMS = [(6, 64, 176, 256, 64, 1, 1, 0), (6, 64, 176, 64, 64, 3, 1, 1), (6, 64, 176, 64, 256, 1, 1, 0), (6, 64, 176, 128, 128, 3, 2, 1),
         (6, 32, 88, 128, 128, 3, 1, 1), (6, 32, 88, 512, 128, 1, 1, 0), (6, 16, 44, 256, 256, 3, 1, 1), (6, 16, 44, 1024, 256, 1, 1, 0),
         (6, 8, 22, 512, 512, 3, 1, 1), (6, 8, 22, 512, 2048, 1, 1, 0), (6, 64, 176, 256, 256, 3, 1, 1), (6, 64, 176, 256, 256, 1, 1, 0),
         (1, 160, 240, 64, 64, 3, 1, 1), (1, 80, 120, 128, 128, 3, 1, 1), (1, 40, 60, 256, 256, 3, 1, 1), (1, 160, 240, 384, 72, 1, 1, 0),
         (1, 160, 240, 512, 256, 3, 1, 1), (1, 160, 240, 256, 256, 3, 1, 1), (1, 160, 240, 1024, 1024, 3, 1, 1), (1, 160, 240, 640, 384, 3, 1, 1),
         (6, 16, 44, 512, 512, 3, 1, 1), (6, 64, 176, 512, 256, 3, 1, 1)]
if os.environ.get("WGRAD_BENCH_3X3_ONLY", "0") == "1":
    GEOMS = [g for g in GEOMS if g[5] == 3 and g[6] == 1]
LIB = os.environ.get("WGRAD_BENCH_LIBRARY", "1") == "1"
if os.environ.get("WGRAD_BENCH_ONE"):                      # "B,H,W,cin,cout,k,s,p": one geometry (counter passes)
    GEOMS = [tuple(int(v) for v in os.environ["WGRAD_BENCH_ONE"].split(","))]
CHAIN = os.environ.get("WGRAD_BENCH_CHAIN", "1") == "1"


def clock(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3 / n
        best = t if best is None else min(best, t)
    return best


for B, H, W, cin, cout, k, s, p in GEOMS:
    x = torch.randn(B, cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    Ho, Wo = (H + 2 * p - (k - 1) - 1) // s + 1, (W + 2 * p - (k - 1) - 1) // s + 1
    g = torch.randn(B, cout, Ho, Wo, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, k, k, device=dev).contiguous(memory_format=torch.channels_last)
    xs, gs = ops.split_f32(x), ops.split_f32(g)
    flops = 2.0 * B * Ho * Wo * cin * cout * k * k
    os.environ["OMNIHD_WGRAD_NHWC"] = "1"
    t_n = clock(lambda: ops.conv_wgrad_split(xs, gs, k, s, p, 1))
    os.environ["OMNIHD_WGRAD_NHWC"] = "0"
    t_c = clock(lambda: ops.conv_wgrad_split(xs, gs, k, s, p, 1)) if CHAIN else float("nan")
    t_m = clock(lambda: torch.ops.aten.convolution_backward(g, x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [False, True, False])[1]) if LIB else float("nan")
    os.environ.pop("OMNIHD_WGRAD_NHWC")
    rule = ops.wgrad_nhwc_preferred(B, H, W, cin, Ho, Wo, cout, k, s, p, 1)
    print(f"{B}x{H}x{W} {cin:4d}->{cout:4d} k{k} s{s} | nhwc {t_n:7.1f} us ({flops/t_n/1e6:5.0f} TF eff) | chain {t_c:7.1f} us | library fp32 {t_m:7.1f} us | rule picks {'nhwc' if rule else 'chain'}", flush=True)

#!/usr/bin/env python3
"""One convolution geometry, N launches of the implicit-GEMM kernel (PMC / kernel-trace workload)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from omnihd_amd import ops
B, H, W, cin, cout, k = (int(v) for v in (sys.argv[1:7] + ["1", "160", "240", "1024", "1024", "3"][len(sys.argv) - 1:]))
tile = int(sys.argv[7]) if len(sys.argv) > 7 else 0
split = len(sys.argv) > 8 and sys.argv[8] == "split"          # the fp32-grade split kernel instead of the bf16 one
dev = torch.device("cuda:0")
if split:
    xf = torch.randn(B, cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    wf = (torch.randn(cout, cin, k, k, device=dev) * 0.02).contiguous(memory_format=torch.channels_last)
    xs, ws = ops.split_f32(xf), ops.split_f32(wf)
    for _ in range(10):
        y = ops.conv_fwd_split(xs, ws, None, 1, tile)
    torch.cuda.synchronize()
    print("ok", float(y.abs().mean()))
    sys.exit(0)
x = torch.randn(B, cin, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
w = (torch.randn(cout, cin, k, k, device=dev) * 0.02).bfloat16().contiguous(memory_format=torch.channels_last)
for _ in range(10):
    y = ops.conv_fwd(x, w, None, 1, tile)
torch.cuda.synchronize()
print("ok", float(y.float().abs().mean()))

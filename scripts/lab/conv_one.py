#!/usr/bin/env python3
"""One convolution geometry, N launches of the implicit-GEMM kernel (PMC / kernel-trace workload)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from omnihd_amd import ops
B, H, W, cin, cout, k = (int(v) for v in (sys.argv[1:7] + ["1", "160", "240", "1024", "1024", "3"][len(sys.argv) - 1:]))
tile = int(sys.argv[7]) if len(sys.argv) > 7 else 0
dev = torch.device("cuda:0")
x = torch.randn(B, cin, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
w = (torch.randn(cout, cin, k, k, device=dev) * 0.02).bfloat16().contiguous(memory_format=torch.channels_last)
for _ in range(10):
    y = ops.conv_fwd(x, w, None, 1, tile)
torch.cuda.synchronize()
print("ok", float(y.float().abs().mean()))

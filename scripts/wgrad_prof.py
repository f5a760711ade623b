import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
from omnihd_amd import ops
for cin, cout in [(1024, 1024), (512, 256)]:
    x = torch.randn(1, cin, 160, 240, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    g = torch.randn(1, cout, 160, 240, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for _ in range(5):
        ops.conv3x3_wgrad(x, g)
torch.cuda.synchronize()

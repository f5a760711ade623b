import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch, bench
wl = bench.BevOps("r1", 1, torch.device("cuda:0"), 1234)
t = bench.time_kernel(wl.pool_fwd, len(wl.sets), 40)
print(f"dbg={os.environ.get('OMNIHD_FWD_DBG','0')} unroll={os.environ.get('OMNIHD_FWD_UNROLL','4')}: {t*1e6:.1f} us")

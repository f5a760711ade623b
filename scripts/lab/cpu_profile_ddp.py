#!/usr/bin/env python3
"""cProfile of the host side of the fp32 step inside a one-rank RCCL group under DistributedDataParallel (what does DDP add?)."""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import bench  # noqa: F401
import torch
import torch.distributed as dist
from omnihd_amd.harness import FusionTrainStep
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29741")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
st = FusionTrainStep(res="r1", batch=1, radar_dims=7, dtype="fp32", ddp=True, miopen_find=True)
for _ in range(6):
    st.step()
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
st.step(); torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    st.step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print("\n".join(l[:170] for l in s.getvalue().splitlines()[:62]))
dist.destroy_process_group()

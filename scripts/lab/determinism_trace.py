#!/usr/bin/env python3
"""Where does run-to-run gradient noise enter the backward pass?  Records the gradient arriving at every module output in
two identical forward/backward passes and lists, in backward order, the modules whose output gradient differs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import bench  # noqa: F401
import torch
from omnihd_amd.harness import FusionTrainStep
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
st = FusionTrainStep(res="r1", batch=1, radar_dims=7, dtype=dt, miopen_find=False)
for _ in range(4):
    st.step()
torch.cuda.synchronize()
m = st.raw_model
store, order = {}, []


def fwd_hook(name):
    def hook(mod, args, out):
        outs = out if isinstance(out, (tuple, list)) else (out,)
        for i, o in enumerate(outs):
            if torch.is_tensor(o) and o.requires_grad:
                key = f"{name}[{i}]"
                if key not in order:
                    order.append(key)
                o.register_hook(lambda g, key=key: store.setdefault(key, []).append(g.detach().float().clone()))
    return hook


for name, mod in m.named_modules():
    if name:
        mod.register_forward_hook(fwd_hook(name))
fwd = {}
for trial in range(2):
    b = st.batches[0]
    st.opt.zero_grad(set_to_none=True)
    torch.manual_seed(123)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dt == "bf16"):
        losses = st.model(return_loss=True, **b)
    total = sum(v if torch.is_tensor(v) else sum(v) for v in losses.values())
    total.backward()
    torch.cuda.synchronize()
print("modules with output grads:", len(order))
shown = 0
for key in reversed(order):
    g = store.get(key)
    if not g or len(g) < 2 or g[0].shape != g[1].shape:
        continue
    d = float((g[0] - g[1]).abs().max() / g[0].abs().max().clamp_min(1e-30))
    flag = "DIFF" if d > 0 else "same"
    if d > 0 or shown < 400:
        print(f"{flag} {d:9.2e}  {key}  {tuple(g[0].shape)}")
        shown += 1

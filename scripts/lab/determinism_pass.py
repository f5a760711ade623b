#!/usr/bin/env python3
"""Is one fp32 training-mode forward + backward of the R1 detector run-to-run identical on the GPU?  Same model object, same batch,
twice: max relative difference of intermediate outputs and gradients, per stage."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from omnihd_amd.harness import FusionTrainStep
os.environ.setdefault("OMNIHD_FP32_CONV", "split")
torch.backends.cudnn.allow_tf32 = False
st = FusionTrainStep(res="r1", batch=1, radar_dims=7, device="cuda:0", seed=5, dtype="fp32", sets=1)
m, b = st.raw_model, st.batches[0]
m.train()
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout):
        mod.p = 0.0
acts = {}
def hook(name):
    def f(mod, inp, out):
        if torch.is_tensor(out):
            acts.setdefault(name, []).append(out.detach().float().clone())
    return f
order = []
for n, mod in m.named_modules():
    if not list(mod.children()) or n in ("reduc_conv",):
        mod.register_forward_hook(hook(n))
runs = []
for it in range(2):
    m.zero_grad(set_to_none=True)
    fd = m.extract_feat(b["points"], img=b["img"], img_metas=b["img_metas"])
    outs = m.pts_bbox_head(fd["pts_feats"])
    losses = m.pts_bbox_head.loss(*outs, b["gt_bboxes_3d"], b["gt_labels_3d"], b["img_metas"])
    dl, _ = m.lift_splat_shot_vis.get_depth_loss(b["img_depth"], fd["depth_dist"], "kld")
    (sum(v[0] for v in losses.values()) + dl).backward()
    torch.cuda.synchronize()
    runs.append(dict(depth=fd["depth_dist"].detach().clone(), bev=fd["pts_feats"][0].detach().clone(),
                     grads={n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}))
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
print("forward, module outputs (run 2 vs run 1):")
shown = 0
for n, v in acts.items():
    if len(v) >= 2 and v[0].shape == v[-1].shape and len(v) % 2 == 0:
        h = len(v) // 2
        same = all(torch.equal(v[i], v[h + i]) for i in range(h))
        if not same and shown < 12:
            shown += 1
            print(f"   FIRST DIFFERENCES: {n:60s} {rel(v[h], v[0]):.2e}  {type(dict(m.named_modules())[n]).__name__} {tuple(v[0].shape)}")
print("   leaf modules with identical outputs:", sum(1 for n, v in acts.items() if len(v) % 2 == 0 and all(torch.equal(v[i], v[len(v)//2 + i]) for i in range(len(v)//2))), "of", len(acts))
print("depth", rel(runs[1]["depth"], runs[0]["depth"]), "bev", rel(runs[1]["bev"], runs[0]["bev"]))
worst = sorted(((rel(runs[1]["grads"][n], g), n) for n, g in runs[0]["grads"].items()), reverse=True)
print("gradients: identical", sum(1 for r, _ in worst if r == 0.0), "of", len(worst))
for r, n in worst[:25]:
    print(f"   {r:.2e}  {n}")

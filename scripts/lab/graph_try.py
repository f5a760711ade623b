#!/usr/bin/env python3
"""Experiment: hipGraph capture (torch.cuda.make_graphed_callables) of static-shape sub-networks of the R1 training step.
python3 scripts/lab/graph_try.py [comma-separated module paths | none] [steps]
Prints the losses of a fixed number of steps (to compare with the eager run) and the step time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import bench  # noqa: F401  (seeds the MIOpen user db)
import torch
from omnihd_amd.harness import FusionTrainStep

names = [n for n in (sys.argv[1] if len(sys.argv) > 1 else "none").split(",") if n and n != "none"]
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
st = FusionTrainStep(res="r1", batch=1, radar_dims=7, dtype="bf16", miopen_find=False)
m = st.raw_model
if os.environ.get("GRAPH_SIDE_STREAM"):
    # everything (eager warm-up, capture, replays) on a non-default stream
    _work_stream = torch.cuda.Stream()
    _work_stream.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(_work_stream)
for _ in range(4):
    st.step()
torch.cuda.synchronize()


def resolve(root, path):
    for p in path.split("."):
        root = getattr(root, p)
    return root


class _Star(torch.nn.Module):
    """inner(list_of_tensors) called as star(*tensors): make_graphed_callables takes tensor arguments only."""

    def __init__(self, inner):
        super().__init__()
        self.inner = inner

    def forward(self, *xs):
        return self.inner(list(xs))


class _Unstar(torch.nn.Module):
    def __init__(self, star):
        super().__init__()
        self.star = star

    def forward(self, xs):
        return self.star(*xs)


def set_module(root, path, new):
    parts = path.split(".")
    for p in parts[:-1]:
        root = getattr(root, p)
    setattr(root, parts[-1], new)


if names:
    # capture the real inputs of each module with a pre-hook during one eager step
    seen, hooks = {}, []
    for n in names:
        mod = resolve(m, n)
        hooks.append(mod.register_forward_pre_hook(lambda mod_, args, n=n: seen.__setitem__(n, args)))
    st.step()
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    samples = {}
    for n in names:
        a = seen[n]
        star = len(a) == 1 and isinstance(a[0], (list, tuple))
        flat = a[0] if star else a
        samples[n] = (star, tuple(t.detach().clone().requires_grad_(t.requires_grad) for t in flat))
    # no autograd graph of an eager step may stay alive: its AccumulateGrad nodes are tied to the default stream and would
    # pull that stream into the capture (observed: segmentation fault in hipStreamEndCapture)
    seen.clear(); del a, flat
    st.last_losses = None
    st.opt.zero_grad(set_to_none=True)
    import gc; gc.collect()
    torch.cuda.synchronize()
    GRAPHED = {}
    for n in names:
        mod = resolve(m, n)
        star, args = samples[n]
        eager_forward = mod.forward
        print("graphing", n, "star" if star else "", [tuple(t.shape) + (str(t.dtype), t.requires_grad) for t in args], flush=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
            if star:
                g = torch.cuda.make_graphed_callables(_Star(mod), args, num_warmup_iters=3)
                set_module(m, n, _Unstar(g))
            else:
                torch.cuda.make_graphed_callables(mod, args, num_warmup_iters=3)
                gf = mod.forward
                if os.environ.get("GRAPH_CLONE_OUT"):
                    def cloned(*a, gf=gf):
                        o = gf(*a)
                        return tuple(t.clone() for t in o) if isinstance(o, tuple) else o.clone()
                    mod.forward = cloned
                GRAPHED[n] = (mod, eager_forward, mod.forward)
        st.opt.zero_grad(set_to_none=True)
        gc.collect()
    torch.cuda.synchronize()

if os.environ.get("GRAPH_CHECK"):
    # the SAME model with the graphed forwards switched on / off: forward losses must agree (the forward pass is
    # deterministic), gradients to within the run-to-run noise of the eager backward (atomics in MIOpen's solvers)
    from omnihd_amd import ops

    def one_pass(trial):
        b = st.batches[trial % 2]
        st.opt.zero_grad(set_to_none=not os.environ.get("GRAPH_KEEP_GRAD"))
        torch.manual_seed(123 + trial)
        with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
            losses_ = st.model(return_loss=True, **b)
        total = sum(v if torch.is_tensor(v) else sum(v) for v in losses_.values())
        total.backward()
        torch.cuda.synchronize()
        return ({k_: float(v_ if torch.is_tensor(v_) else sum(v_)) for k_, v_ in losses_.items()},
                {n_: p_.grad.detach().float().clone() for n_, p_ in m.named_parameters() if p_.grad is not None})

    def use(which):
        for n in names:
            mod = GRAPHED[n][0]
            mod.forward = GRAPHED[n][1 if which == "eager" else 2]

    if os.environ.get("GRAPH_SEQ"):
        keep = []
        for i_, ch in enumerate(os.environ["GRAPH_SEQ"]):
            # G / E: graphed / eager pass on frame 0;  g / e: on alternating frames;  S: graphed pass + optimiser step;
            # K: graphed pass whose gradient copies are kept alive
            if ch == "F":                       # eager forward only, graph dropped without a backward
                use("eager")
                with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
                    tmp_ = st.model(return_loss=True, **st.batches[0])
                del tmp_; torch.cuda.synchronize(); print("F (eager forward only)"); continue
            if ch == "B":                       # eager forward + backward of the image backbone alone
                use("eager")
                with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
                    outs_ = m.img_backbone(st.batches[0]["img"].view(-1, 3, 256, 704).contiguous(memory_format=torch.channels_last))
                sum(o_.float().mean() for o_ in outs_).backward()
                del outs_; torch.cuda.synchronize(); print("B (eager backbone fwd+bwd only)"); continue
            if ch == "A":                       # allocator churn only
                tmp_ = [torch.empty(1 << 30, dtype=torch.uint8, device="cuda") for _ in range(24)]
                del tmp_; torch.cuda.synchronize(); print("A (allocate / free 24 GiB)"); continue
            use("eager" if ch in "Ee" else "graph")
            l_, g_ = one_pass(i_ if ch in "geS" else 0)
            if ch == "K":
                keep.append(g_)
            bad = [n_ for n_, t_ in g_.items() if not bool(torch.isfinite(t_).all())]
            if 'ref_' not in globals():
                ref_ = g_
            dev_ = sorted(((float((g_[n_] - ref_[n_]).abs().max() / ref_[n_].abs().max().clamp_min(1e-20)), n_) for n_ in ref_
                           if bool(torch.isfinite(g_[n_]).all())), reverse=True)
            print(ch, "loss", round(sum(l_.values()), 5), "non-finite grads:", len(bad), bad[:3],
                  "| worst deviation from the first pass:", [(f"{d_:.2e}", n_) for d_, n_ in dev_[:3]],
                  "| params off by > 10x:", sum(1 for d_, _ in dev_ if d_ > 10), flush=True)
            if ch == "S":
                names_ = {id(p_): n_ for n_, p_ in m.named_parameters()}
                for rep in range(2):
                    norms = torch._foreach_norm([p_.grad for p_ in st.params], 2.0)
                    badn = [(names_[id(p_)], float(v_), float(p_.grad.abs().max()), float(g_[names_[id(p_)]].abs().max()),
                             float(torch.linalg.vector_norm(p_.grad.float())), p_.grad.dtype, p_.grad.is_contiguous(), tuple(p_.grad.stride()))
                            for p_, v_ in zip(st.params, norms) if not bool(torch.isfinite(v_))]
                    print("   foreach_norm non-finite:", len(badn), badn[:3], flush=True)
                tn = torch.nn.utils.clip_grad_norm_(st.params, max_norm=35, norm_type=2)
                st.opt.step(); ops.refresh_bf16_shadows()
                print("   step: total norm", float(tn), flush=True)
        sys.exit(0)
    for trial in range(3):
        use("graph"); lg, gg = one_pass(trial)
        use("eager"); le, ge = one_pass(trial)
        le2, ge2 = one_pass(trial)
        print(f"trial {trial}: losses graphed {lg}\n          eager   {le}")
        rel = lambda a, b_: float((a - b_).abs().max() / b_.abs().max().clamp_min(1e-20))
        rows = sorted(((rel(gg[n_], ge[n_]), rel(ge2[n_], ge[n_]), n_) for n_ in ge), reverse=True)
        print("   only graphed:", [n_ for n_ in gg if n_ not in ge][:10], " only eager:", [n_ for n_ in ge if n_ not in gg][:10])
        print("   non-finite graphed grads:", [n_ for n_, g_ in gg.items() if not bool(torch.isfinite(g_).all())][:10])
        gn = lambda d_: float(torch.sqrt(sum((g_ ** 2).sum() for g_ in d_.values())))
        print(f"   grad norm graphed {gn(gg):.5f} eager {gn(ge):.5f} eager again {gn(ge2):.5f}")
        print("   worst grad deviations graph-vs-eager (eager-vs-eager beside):")
        for d, d0, n_ in rows[:8]:
            print(f"     {d:9.2e} ({d0:9.2e})  {n_}")
        use("graph")
        l4, g4 = one_pass(trial)
        print("   4th pass (graph) losses", l4, "non-finite grads:", [n_ for n_, g_ in g4.items() if not bool(torch.isfinite(g_).all())][:6])
        print("   live p.grad non-finite:", [n_ for n_, p_ in m.named_parameters() if p_.grad is not None and not bool(torch.isfinite(p_.grad).all())][:6])
        norms = torch._foreach_norm([p_.grad for p_ in st.params], 2.0)
        print("   foreach norms non-finite at:", [i_ for i_, v_ in enumerate(norms) if not bool(torch.isfinite(v_))][:10], "of", len(norms),
              " stacked norm", float(torch.linalg.vector_norm(torch.stack(norms), 2.0)))
        tn = torch.nn.utils.clip_grad_norm_(st.params, max_norm=35, norm_type=2)
        print("   clip_grad_norm_ total norm", float(tn), " params with grad None:", sum(1 for p_ in st.params if p_.grad is None))
        st.opt.step(); ops.refresh_bf16_shadows()
        bad = [n_ for n_, p_ in m.named_parameters() if not bool(torch.isfinite(p_).all())]
        print("   non-finite params after the step:", len(bad), bad[:8])
    sys.exit(0)

# fresh optimiser state does not matter for the comparison: both runs did the same number of steps before
losses = []
if names:
    import contextlib
    orig = torch.autocast

    def no_cache_autocast(*a, **k):
        k.setdefault("cache_enabled", False)
        return orig(*a, **k)
    torch.autocast = no_cache_autocast
for _ in range(3):
    losses.append(float(st.step()))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n_steps):
    st.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n_steps * 1e3
losses.append(float(st.step()))
gn = float(torch.sqrt(sum((p.grad.float() ** 2).sum() for p in st.params if p.grad is not None)))
print(f"RESULT graphed={names or 'none'} ms_per_step={dt:.2f} losses={['%.5f' % v for v in losses]} grad_norm={gn:.5f}", flush=True)

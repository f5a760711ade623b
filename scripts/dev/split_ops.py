"""One-time tool (round 6): cut omnihd_amd/ops.py (3 100 lines) into the package omnihd_amd/ops/ by concern.  Top-level statements keep
their text (comments included); each goes to a module by name or by the section it stood in; cross-module names are imported
explicitly; the package's __init__ re-exports every top-level name, private ones included, so `ops.X` means what it meant.
Usage: python scripts/dev/split_ops.py <ops.py> <out_dir>"""
import ast
import builtins
import dis
import os
import sys

SRC, OUT = sys.argv[1], sys.argv[2]
text = open(SRC).read()
lines = text.split("\n")
tree = ast.parse(text)

# ---- module of a top-level statement: by name first, else by the line it starts on -------------------------------------------------
BY_NAME = {
    "_core": ["_ptr", "_current_device", "_raw_stream", "_stream", "_want", "_same_device", "_workspace", "_NULL_CTX", "_SIZE_CACHE", "_on",
              "_WGRAD_WS", "_wgrad_workspace", "_want_cl", "_pair_same", "deterministic", "FAST_PATHS", "_rows_view", "_f32c", "_CL"],
    "conv_kernels": ["conv_fwd_split", "conv_split_geometry", "column_sums", "conv_fwd_f16", "conv_wgrad_f16"],
    "planes": ["cast_f16", "_AMAX_RING", "_AMAX_SLOTS", "_amax_slot", "_f16_plane"],
    "policy": ["f16_handover"],
    "weights": ["f16_weight", "refresh_f16_shadows", "_F16_SHADOW"],
    "conv_fp32": ["conv_split_supported", "conv_split"],
    "radar": ["radar_merge"],
}
NAME_TO_MOD = {n: m for m, ns in BY_NAME.items() for n in ns}
RANGES = [  # (first line of the section, module) — ascending
    (1, "_core"), (66, "pool"), (472, "radar"), (710, "conv_kernels"), (957, "policy"), (1154, "weights"), (1303, "policy"),
    (1390, "planes"), (1514, "weights"), (1628, "conv_kernels"), (1668, "policy"), (1720, "conv_fp32"), (1882, "streams"),
    (2118, "conv_fp32"), (2399, "conv_bf16"), (2677, "misc"), (2764, "norm"), (3011, "misc"), (3080, "radar"),
]
ORDER = ["_core", "pool", "radar", "conv_kernels", "policy", "planes", "weights", "streams", "conv_fp32", "conv_bf16", "norm", "misc"]
DOC = {
    "_core": "Shared plumbing of the operator wrappers: raw stream handle, device guard, argument checks, workspaces, live counters.",
    "pool": "bev_pool_v2 / bev_pool (v1) wrappers, the depth-head epilogue and rank preparation (csrc/bev_pool_v2.hip, bev_pool_v1.hip, rank_prep.hip, depth_head.hip).",
    "radar": "Radar branch: hard voxelisation, pillar scatter, fused pillar feature net, sweep merge (csrc/voxelize.hip, pillar_scatter.hip, pillar_pfn.hip, radar_merge.hip).",
    "conv_kernels": "Convolution kernels behind the C ABI: weight gradients, implicit-GEMM forward / data gradient (bf16, split, half), the general strided kernel, column sums.",
    "policy": "Which implementation runs a convolution: the persisted / measured choice tables and the environment policies (OMNIHD_CONV_POLICY, OMNIHD_FP32_CONV).",
    "planes": "Operand planes of fp32 activations: the hi / lo bf16 split, the IEEE-half cast with its device-side scale, and the producer -> consumer hand-over tags.",
    "weights": "Weight images (bf16, split, half; forward and data-gradient layouts) kept current behind the optimiser step with one launch.",
    "streams": "Weight gradients on a side stream (plain and under DistributedDataParallel bucket views) and the live fast-path report.",
    "conv_fp32": "The fp32 step's convolutions as autograd Functions: fp32-grade split form (_ConvSplit) and TF32-grade half form (_ConvF16).",
    "conv_bf16": "The bf16 step's convolutions and transposed convolutions as autograd Functions, bias gradients by column sums.",
    "norm": "BatchNorm epilogues as autograd Functions: frozen (affine + residual + ReLU) and training mode (csrc/affine_act.hip, batch_norm.hip).",
    "misc": "Deformable-convolution sampling, rotated NMS, fused anchor targets + detection losses (csrc/dcn_sample.hip, nms_rotated.hip, anchor_loss.hip).",
}


def defined_names(node):
    out = []
    if isinstance(node, (ast.FunctionDef, ast.ClassDef)):
        out.append(node.name)
    elif isinstance(node, ast.Assign):
        for t in node.targets:
            for n in ast.walk(t):
                if isinstance(n, ast.Name):
                    out.append(n.id)
    elif isinstance(node, ast.AnnAssign) and isinstance(node.target, ast.Name):
        out.append(node.target.id)
    elif isinstance(node, (ast.If, ast.Try)):
        for sub in ast.iter_child_nodes(node):
            if isinstance(sub, ast.stmt):
                out += defined_names(sub)
    return out


body = [n for n in tree.body]
header_end = 0
chunks = []          # (module, first_line, last_line, names)
prev_end = 0
for i, node in enumerate(body):
    start = min([node.lineno] + [d.lineno for d in getattr(node, "decorator_list", [])])
    end = node.end_lineno
    if isinstance(node, (ast.Import, ast.ImportFrom)) or (isinstance(node, ast.Expr) and i == 0):
        prev_end = end
        continue
    first = prev_end + 1                         # leading blank lines / comments travel with the statement
    names = defined_names(node)
    mod = next((NAME_TO_MOD[n] for n in names if n in NAME_TO_MOD), None)
    if mod is None:
        mod = [m for s, m in RANGES if s <= start][-1]
    chunks.append((mod, first, end, names))
    prev_end = end
tail = "\n".join(lines[prev_end:]).strip()
assert not tail, tail[:200]

owner = {}
for mod, _, _, names in chunks:
    for n in names:
        owner.setdefault(n, mod)

HEADER = '''import contextlib
import ctypes
import os
import weakref

import numpy as np
import torch

from .._env import env as _env
from .._lib import check, lib
'''


def module_source(mod, imports=""):
    parts = [f'"""{DOC[mod]}\n(Part of omnihd_amd.ops — the tensor-level wrappers over the C ABI; `from omnihd_amd import ops` exposes every name.)"""\n' + HEADER + imports]
    for m, a, b, _ in chunks:
        if m == mod:
            seg = "\n".join(lines[a - 1:b])
            seg = seg.replace("from . import plan as _plan", "from .. import plan as _plan")
            parts.append(seg)
    return "\n".join(parts).rstrip() + "\n"


def global_loads(src, name):
    code = compile(src, name, "exec")
    seen, stack = set(), [code]
    while stack:
        c = stack.pop()
        for ins in dis.get_instructions(c):
            if ins.opname in ("LOAD_GLOBAL", "LOAD_NAME"):
                seen.add(ins.argval)
        stack += [k for k in c.co_consts if hasattr(k, "co_code")]
    return seen


os.makedirs(OUT, exist_ok=True)
deps = {}
for mod in ORDER:
    used = global_loads(module_source(mod), mod)
    need = {}
    for n in sorted(used):
        o = owner.get(n)
        if o is not None and o != mod:
            need.setdefault(o, []).append(n)
    deps[mod] = need
    imp = "".join(f"from .{o} import {', '.join(ns)}\n" if len(', '.join(ns)) < 110 else
                  f"from .{o} import ({', '.join(ns)})\n" for o, ns in sorted(need.items(), key=lambda kv: ORDER.index(kv[0])))
    open(os.path.join(OUT, mod + ".py"), "w").write(module_source(mod, imp))
    # anything loaded that nobody defines?
    known = set(owner) | set(dir(builtins)) | {"contextlib", "ctypes", "os", "weakref", "np", "torch", "_env", "check", "lib", "__name__", "__file__"}
    missing = sorted(n for n in used if n not in known)
    if missing:
        print(f"[{mod}] UNDEFINED: {missing}")

for mod in ORDER:
    later = [o for o in deps[mod] if ORDER.index(o) > ORDER.index(mod)]
    print(f"{mod:13s} <- {sorted(deps[mod])}" + (f"   !! imports later modules {later}" if later else ""))

init = ['"""Tensor-level wrappers over the C ABI (argument checks + pointer / stream hand-over only), by concern:\n' +
        "".join(f"  {m:13s} {DOC[m]}\n" for m in ORDER) +
        'Every top-level name of every part is re-exported here, private ones included: `ops.X` is the public face (tests spy on and\n'
        'replace functions through it); state that code REBINDS lives in one part and is reached through that part."""\n']
for mod in ORDER:
    names = [n for m, _, _, ns in chunks if m == mod for n in ns]
    names = list(dict.fromkeys(names))
    init.append(f"from . import {mod}\n")
    buf, line = [], "from .%s import (" % mod
    for n in names:
        if len(line) + len(n) + 2 > 128:
            buf.append(line.rstrip())
            line = "    "
        line += n + ", "
    buf.append(line.rstrip().rstrip(",") + ")")
    init.append("\n".join(buf) + "\n")
init.append("\nPARTS = (" + ", ".join(ORDER) + ")\n")
open(os.path.join(OUT, "__init__.py"), "w").write("".join(init))
print("lines:", {m: module_source(m).count("\n") for m in ORDER})

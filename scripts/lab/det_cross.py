#!/usr/bin/env python3
"""Cross-process determinism diagnosis: the child of tests/test_determinism_gpu.py under several switch settings, every pair of
runs compared tensor by tensor.  python3 scripts/lab/det_cross.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_determinism_gpu as T

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
if os.environ.get("DET_SEED_DB", "1") == "1":      # as tests/conftest.py's `cuda` fixture does: the children inherit MIOPEN_USER_DB_PATH
    sys.path.insert(0, os.path.join(ROOT, "omnihd-scenes_amd"))
    from omnihd_amd.harness import seed_miopen_db
    print("seeded MIOpen user db:", seed_miopen_db())
VARIANTS = [
    ("default", {}),
    ("default again", {}),
    ("in line (no side streams)", {"OMNIHD_DUAL_STREAM": "0", "OMNIHD_WGRAD_OVERLAP": "0"}),
    ("in line again", {"OMNIHD_DUAL_STREAM": "0", "OMNIHD_WGRAD_OVERLAP": "0"}),
    ("no wgrad overlap", {"OMNIHD_WGRAD_OVERLAP": "0"}),
    ("no dual stream", {"OMNIHD_DUAL_STREAM": "0"}),
    ("sort voxeliser", {"OMNIHD_VOXELIZE_GRID": "0"}),
    ("no nhwc wgrad", {"OMNIHD_WGRAD_NHWC": "0"}),
    ("late radar join", {"OMNIHD_RADAR_JOIN": "late"}),
    ("torch anchor loss", {"OMNIHD_ANCHOR_LOSS": "0"}),
]
only = os.environ.get("DET_VARIANTS")
runs = []
for name, env in VARIANTS:
    if only and name.split()[0] not in only.split(","):
        continue
    (digest, losses), per = T._run("r1", steps, env)
    runs.append((name, digest, losses, per))
    print(f"{name:28s} {digest[:12]} losses {losses}", flush=True)
ref = runs[0]
for name, digest, losses, per in runs[1:]:
    diff = [k for k in per if per[k] != ref[3].get(k)]
    print(f"\n== {name} vs {ref[0]}: {len(diff)} of {len(per)} tensors differ")
    by = {}
    for k in diff:
        kind = k.split(" ")[0]
        top = k.split(" ")[-1].split(".")[0]
        by.setdefault((kind, top), []).append(k)
    for (kind, top), ks in by.items():
        print(f"   {kind:8s} {top:24s} {len(ks):4d}   e.g. {ks[0]}")

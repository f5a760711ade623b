#!/usr/bin/env python3
"""Condense rocprofv3 output directories (kernel stats + PMC csv) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for f in glob.glob(os.path.join(out, "kt", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, out))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print(f"  {r.get('Name','')[:90]:90s} calls {r.get('Calls','')} avg_ns {r.get('AverageNs','')} total_ns {r.get('TotalDurationNs','')} pct {r.get('Percentage','')}")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if "k_pool" not in name:
                continue
            short = "fwd_tiles" if "fwd_tiles" in name else ("bwd" if "bwd" in name else name[:30])
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                print(f"  {k:10s} {c:32s} mean/launch {sum(v)/len(v):.4g}  (n={len(v)})")

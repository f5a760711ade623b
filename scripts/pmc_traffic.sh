#!/bin/bash
# HBM traffic of the pooling forward from PMC counters, calibrated on a copy of known size.
OUT=${1:-gpurun_out/pmc_traffic}; export TMPDIR=/tmp; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --output-format csv --pmc $C -d $OUT/$C -o pmc -- python3 scripts/calibrate_pmc.py > $OUT/$C.log 2>&1
done
python3 - <<PY
import csv, glob, json
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True)[0]
    copy, pool = [], []
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        n = r["Kernel_Name"]
        if ("k_pool_fwd_tiles" in n or "k_pool_fwd_lean" in n): pool.append(float(r["Counter_Value"]))
        elif "MulFunctor" in n or "mul" in n.lower(): copy.append(float(r["Counter_Value"]))
    vals[c] = (sum(copy[-8:]) / max(len(copy[-8:]), 1), sum(pool) / max(len(pool), 1))
copy_bytes = 128 * 1024 * 1024
res = {"copy_bytes": copy_bytes}
for c in vals:
    kb_copy, kb_pool = vals[c]
    factor = copy_bytes / (kb_copy * 1024) if kb_copy else None
    res[c] = {"copy_raw_kb": kb_copy, "pool_raw_kb": kb_pool, "calibration_factor": factor,
              "pool_bytes_corrected": kb_pool * 1024 * factor if factor else None}
if res["FETCH_SIZE"]["pool_bytes_corrected"] and res["WRITE_SIZE"]["pool_bytes_corrected"]:
    res["hbm_bytes_per_launch"] = res["FETCH_SIZE"]["pool_bytes_corrected"] + res["WRITE_SIZE"]["pool_bytes_corrected"]
print(json.dumps(res, indent=1))
json.dump(res, open("$OUT/pmc_bev_pool_fwd.json", "w"), indent=1)
PY
find $OUT -type f -size +1M -delete

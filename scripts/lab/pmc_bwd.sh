#!/bin/bash
# HBM bytes and L2 hit rate of the pooling kernels from PMC counters (separate passes; FETCH_SIZE calibrated on a copy).
OUT=${1:-gpurun_out/pmc_bwd}; RES=${2:-r1}; export TMPDIR=/tmp; mkdir -p $OUT
timeout 240 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/p1 -o pmc -- python3 scripts/lab/pmc_bwd.py $RES > $OUT/p1.log 2>&1
timeout 240 rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/p2 -o pmc -- python3 scripts/lab/pmc_bwd.py $RES > $OUT/p2.log 2>&1
timeout 240 rocprofv3 --output-format csv --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $OUT/p3 -o pmc -- python3 scripts/lab/pmc_bwd.py $RES > $OUT/p3.log 2>&1
python3 - <<PY
import csv, glob, json, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list)); names = set()
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_pool_fwd" in n: names.add(re.search(r"k_pool_fwd_\w+", n).group(0))
        k = "patch_bwd" if "k_pool_bwd_patch" in n else "fwd_lean" if ("k_pool_fwd_lean" in n or "k_pool_fwd_direct" in n) else "copy" if "MulFunctor" in n or "mul" in n.lower() else None
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v[-6:]) / len(v[-6:]) for c, v in d.items()} for k, d in acc.items()}
cal = 128 * 1024 * 1024 / (res["copy"]["FETCH_SIZE"] * 1024) if "copy" in res and res["copy"].get("FETCH_SIZE") else None
for k, d in res.items():
    if "FETCH_SIZE" in d and cal: d["read_bytes_corrected"] = d["FETCH_SIZE"] * 1024 * cal
    if "WRITE_SIZE" in d: d["write_bytes"] = d["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in d: d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    if "TCP_TCC_READ_REQ_sum" in d: d["l1_miss_share"] = d["TCP_TCC_READ_REQ_sum"] / max(d["TCP_TOTAL_CACHE_ACCESSES_sum"], 1)
res["fetch_calibration_factor"] = cal
res["fwd_kernel"], res["bwd_kernel"] = "+".join(sorted(names)) or "?", "k_pool_bwd_patch"
import hashlib
res["pool_source_sha256"] = hashlib.sha256(open("omnihd-scenes_amd/csrc/bev_pool_v2.hip", "rb").read()).hexdigest()
res["how"] = ("scripts/lab/pmc_bwd.sh: three separate rocprofv3 --pmc passes over scripts/lab/pmc_bwd.py (cold launches: a 128 MiB copy "
              "between them, rotating buffer sets; rows without points keep their zeros); FETCH_SIZE x 1024 x the factor calibrated on the copy")
print(json.dumps(res, indent=1))
json.dump(res, open("$OUT/pmc_pool_$RES.json", "w"), indent=1)
PY
find $OUT -type f -size +1M -delete

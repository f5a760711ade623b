#!/bin/bash
# SQ / TCC counters and kernel trace of the split weight gradient (separate passes).  usage: pmc_wgrad.sh OUT [B H W cin cout k]
OUT=${1:-gpurun_out/pmc_wgrad}; ARGS="${@:2}"; export TMPDIR=/tmp; mkdir -p $OUT
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $OUT/p1 -o pmc -- python3 scripts/lab/wgrad_one.py $ARGS > $OUT/p1.log 2>&1
rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum GRBM_GUI_ACTIVE -d $OUT/p2 -o pmc -- python3 scripts/lab/wgrad_one.py $ARGS > $OUT/p2.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/p4 -o kt -- python3 scripts/lab/wgrad_one.py $ARGS > $OUT/p4.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p[12]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_wgrad_split3" in r["Kernel_Name"] or "k_wgrad_shift" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:34s} {sum(v[-5:])/len(v[-5:]):16.0f}")
for f in glob.glob("$OUT/p4/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "omnihd" in r["Name"]: print("kernel_stats", r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"])
PY
find $OUT -type f -size +1M -delete

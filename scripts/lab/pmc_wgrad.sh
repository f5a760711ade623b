#!/bin/bash
# SQ / TCC counters of the NHWC weight-gradient kernel on one geometry (separate passes).  usage: pmc_wgrad.sh OUT "B,H,W,cin,cout,k,s,p"
OUT=${1:-gpurun_out/pmc_wgrad}; export WGRAD_BENCH_ONE=${2:-1,160,240,1024,1024,3,1,1}; export WGRAD_BENCH_LIBRARY=0 WGRAD_BENCH_CHAIN=0 TMPDIR=/tmp; mkdir -p $OUT
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $OUT/p1 -o pmc -- python3 scripts/lab/wgrad_nhwc_bench.py > $OUT/p1.log 2>&1
rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -d $OUT/p2 -o pmc -- python3 scripts/lab/wgrad_nhwc_bench.py > $OUT/p2.log 2>&1
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d $OUT/p3 -o pmc -- python3 scripts/lab/wgrad_nhwc_bench.py > $OUT/p3.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/p4 -o kt -- python3 scripts/lab/wgrad_nhwc_bench.py > $OUT/p4.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p[123]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_wgrad_nhwc" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("geometry $WGRAD_BENCH_ONE: counters of k_wgrad_nhwc, mean of the last 5 launches")
for k, v in sorted(acc.items()):
    print(f"{k:34s} {sum(v[-5:])/len(v[-5:]):16.0f}")
for f in glob.glob("$OUT/p4/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad" in r["Name"] or "sum_slabs" in r["Name"]: print("kernel_stats", r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"])
PY
tail -n 2 $OUT/p1.log $OUT/p2.log $OUT/p3.log | cut -c1-200
find $OUT -type f -size +1M -delete

#!/bin/bash
# SQ / TCC counters of the implicit-GEMM convolution kernel (separate passes).
OUT=${1:-gpurun_out/pmc_conv}; ARGS="${@:2}"; export TMPDIR=/tmp; mkdir -p $OUT   # usage: pmc_conv.sh OUT [B H W cin cout k tile [split]]
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $OUT/p1 -o pmc -- python3 scripts/lab/conv_one.py $ARGS > $OUT/p1.log 2>&1
rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -d $OUT/p2 -o pmc -- python3 scripts/lab/conv_one.py $ARGS > $OUT/p2.log 2>&1
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d $OUT/p3 -o pmc -- python3 scripts/lab/conv_one.py $ARGS > $OUT/p3.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/p4 -o kt -- python3 scripts/lab/conv_one.py $ARGS > $OUT/p4.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p[123]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_conv_igemm" in r["Kernel_Name"]:  # either kernel of csrc/conv_igemm.hip
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:34s} {sum(v[-5:])/len(v[-5:]):16.0f}")
for f in glob.glob("$OUT/p4/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "igemm" in r["Name"]: print("kernel_stats", r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"])
PY
tail -n 3 $OUT/p1.log $OUT/p2.log $OUT/p3.log | cut -c1-200
find $OUT -type f -size +1M -delete

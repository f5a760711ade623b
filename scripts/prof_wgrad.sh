#!/bin/bash
OUT=${1:-gpurun_out/prof_wgrad}; export TMPDIR=/tmp; mkdir -p $OUT
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
         ${WGRAD_PMC_MORE:+"SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"} \
         ${WGRAD_PMC_MORE:+"TCC_HIT_sum TCC_MISS_sum"} ${WGRAD_PMC_MORE:+"FETCH_SIZE"}; do
  T=$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --output-format csv --pmc $C -d $OUT/p_$T -o pmc -- python3 scripts/wgrad_prof.py > $OUT/p_$T.log 2>&1
done
python3 - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(list))
for f in glob.glob("$OUT/p_*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_wgrad_mfma" in r["Kernel_Name"]:
            acc[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g,cs in acc.items():
    print("grid",g)
    for c,v in sorted(cs.items()): print("   %-34s %.4g"%(c,sum(v)/len(v)))
PY
find $OUT -type f -size +1M -delete

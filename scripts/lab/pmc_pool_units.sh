#!/bin/bash
# Which unit do the pooling kernels keep busy?  Extra rocprofv3 --pmc passes over the workload of pmc_bwd.sh (scripts/lab/pmc_bwd.py):
# texture-address unit (TA: the front of the vector-L1 path) busy cycles, wave cycles by state, instruction mix.  (Every pass under
# its own `timeout`: a pass with the TA_*_STALLED_* counters made rocprofv3 abort and hang on this image — they are not collected.)
OUT=${1:-gpurun_out/pmc_units}; RES=${2:-r1}; export TMPDIR=/tmp; mkdir -p $OUT
timeout 240 rocprofv3 --output-format csv --pmc TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE -d $OUT/u1 -o pmc -- python3 scripts/lab/pmc_bwd.py $RES > $OUT/u1.log 2>&1
timeout 240 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU -d $OUT/u3 -o pmc -- python3 scripts/lab/pmc_bwd.py $RES > $OUT/u3.log 2>&1
python3 - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/u*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "bwd" if "k_pool_bwd_patch" in n else "fwd" if ("k_pool_fwd_lean" in n or "k_pool_fwd_direct" in n) else "copy" if "MulFunctor" in n or "mul" in n.lower() else None
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v[-6:]) / len(v[-6:]) for c, v in d.items()} for k, d in acc.items()}
for k, d in res.items():
    if "TA_BUSY_avr" in d and "GRBM_GUI_ACTIVE" in d:
        d["ta_busy_frac_avg"] = d["TA_BUSY_avr"] / d["GRBM_GUI_ACTIVE"]
        d["ta_busy_frac_max"] = d["TA_BUSY_max"] / d["GRBM_GUI_ACTIVE"]
    if "SQ_WAVE_CYCLES" in d:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM"):
            if c in d: d[c.lower() + "_per_wave_cycle"] = d[c] / d["SQ_WAVE_CYCLES"]
res["how"] = "scripts/lab/pmc_pool_units.sh: separate rocprofv3 --pmc passes over scripts/lab/pmc_bwd.py (mean of the last 6 launches of each kernel)"
print(json.dumps(res, indent=1))
json.dump(res, open("$OUT/pmc_units_$RES.json", "w"), indent=1)
PY
find $OUT -type f -size +1M -delete

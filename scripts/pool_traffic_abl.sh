#!/bin/bash
# Where does the read traffic of the pooling forward come from?  FETCH_SIZE of three ablation builds of the
# library (built on the CPU box into scripts/micro/abl/, see the macro in csrc/bev_pool_v2.hip).
export TMPDIR=/tmp
OUT=gpurun_out/pool_abl; mkdir -p $OUT
for A in 0 1 2 3; do
  LIB=$PWD/scripts/micro/abl/libomnihd_abl$A.so
  [ -f $LIB ] || continue
  export OMNIHD_LIB_PATH=$LIB
  rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/a$A -o pmc -- python3 scripts/calibrate_pmc.py > $OUT/a$A.log 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$OUT/a$A/**/*counter_collection.csv", recursive=True)[0]
pool, copy = [], []
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "FETCH_SIZE": continue
    (pool if ("k_pool_fwd_tiles" in r["Kernel_Name"] or "k_pool_fwd_lean" in r["Kernel_Name"]) else copy if "MulFunctor" in r["Kernel_Name"] else []).append(float(r["Counter_Value"]))
cal = 128 * 1024 * 1024 / (sum(copy[-8:]) / len(copy[-8:]) * 1024)
print("abl $A: fetch %.1f MB per launch (calibration x%.3f)" % (sum(pool) / len(pool) * 1024 * cal / 1e6, cal))
PY
  find $OUT/a$A -type f -size +1M -delete
done

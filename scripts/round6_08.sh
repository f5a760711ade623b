#!/bin/bash
# Round 6: pooling tests after the prune + this round's PMC passes of the pooling kernels (R1, R2)
export TMPDIR=/tmp; out=gpurun_out/r6_08; mkdir -p $out
timeout 1500 python3 -m pytest tests/test_bev_pool_gpu.py tests/test_pool_properties_gpu.py tests/test_device_plan_gpu.py tests/test_prepare_gpu.py tests/test_depth_head_gpu.py tests/test_lss_plain_gpu.py -q > $out/pytest_pool.txt 2>&1; echo "pytest rc $?"; tail -6 $out/pytest_pool.txt
bash scripts/lab/pmc_bwd.sh $out/pmc_r1 r1 > $out/pmc_r1.log 2>&1; cp $out/pmc_r1/pmc_pool_r1.json $out/ 2>/dev/null
bash scripts/lab/pmc_bwd.sh $out/pmc_r2 r2 > $out/pmc_r2.log 2>&1; cp $out/pmc_r2/pmc_pool_r2.json $out/ 2>/dev/null
bash scripts/lab/pmc_pool_units.sh $out/units_r1 r1 > $out/units_r1.log 2>&1; cp $out/units_r1/pmc_units_r1.json $out/ 2>/dev/null
bash scripts/lab/pmc_pool_units.sh $out/units_r2 r2 > $out/units_r2.log 2>&1; cp $out/units_r2/pmc_units_r2.json $out/ 2>/dev/null
python3 - <<PY
import json
for r in ("r1", "r2"):
    try:
        d = json.load(open("$out/pmc_pool_%s.json" % r))
        for k in ("fwd_lean", "patch_bwd"):
            print(r, k, {x: round(d[k][x]) if d[k][x] > 10 else round(d[k][x], 4) for x in ("read_bytes_corrected", "write_bytes", "l2_hit_rate", "l1_miss_share") if x in d[k]})
    except Exception as e:
        print(r, "no pmc", e)
PY
find $out -name "*.csv" -size +1M -delete

#!/bin/bash
# rocprofv3 passes for the pooling kernels; counters in separate runs (no trace domains mixed in).
# usage: scripts/prof_pmc.sh <outdir> [r1|r2]
set -u
OUT=${1:-gpurun_out/prof}; RES=${2:-r1}
export TMPDIR=/tmp
mkdir -p "$OUT"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/kt" -o kt -- python3 scripts/prof_pool.py $RES 20 > "$OUT/kt.log" 2>&1
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "GRBM_GUI_ACTIVE"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --output-format csv --pmc $C -d "$OUT/pmc_$T" -o pmc -- python3 scripts/prof_pool.py $RES 8 > "$OUT/pmc_$T.log" 2>&1
done
python3 scripts/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep only small artefacts (gpurun copies back at most 64 MiB)
find "$OUT" -type f -size +2M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"

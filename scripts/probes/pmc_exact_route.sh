#!/bin/bash
# PMC passes over one exact-route call at the sigma_mlp shape (SiLU-gated columns by default): the remainder kernels' counters.
#   bash scripts/probes/pmc_exact_route.sh <tag> [gaussian|silu_gated]   -> gpurun_out/<tag>_exact_route_pmc.csv
export TMPDIR=/tmp
R=$PWD
TAG=${1:-r04}
FAM=${2:-silu_gated}
: > gpurun_out/${TAG}_exact_route_pmc.csv
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD"; do
  OUT=/tmp/${TAG}_pmcx
  rm -rf $OUT
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass -d $OUT -o p -- python3 $R/scripts/probes/exact_route_timing.py 14336 32768 $FAM exact > $R/gpurun_out/${TAG}_pmcx.log 2>&1) || { echo "PMC pass $pass failed"; tail -5 gpurun_out/${TAG}_pmcx.log; exit 1; }
  DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
  echo "# pass: $pass" >> gpurun_out/${TAG}_exact_route_pmc.csv
  python3 scripts/rocpd_summary.py $DB bygrid | grep -i "i8_lo_\|i8_extract\|i8_copy\|^kernel" | cut -c1-200 >> gpurun_out/${TAG}_exact_route_pmc.csv
  rm -rf $OUT
done
cat gpurun_out/${TAG}_exact_route_pmc.csv

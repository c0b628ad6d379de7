#!/bin/bash
# Per-kernel times of one int8 covariance call on the exact route at the sigma_mlp shape, one data family per trace.
#   bash scripts/probes/prof_exact_route.sh <tag>    -> gpurun_out/<tag>_exact_route_kernels_{gaussian,silu_gated}.csv
export TMPDIR=/tmp
R=$PWD
TAG=${1:-r04}
for fam in gaussian silu_gated; do
  OUT=/tmp/${TAG}_exr
  rm -rf $OUT
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT -o p -- python3 $R/scripts/probes/exact_route_timing.py 14336 32768 $fam exact > $R/gpurun_out/${TAG}_exr_$fam.log 2>&1) || { echo "trace failed"; tail -5 gpurun_out/${TAG}_exr_$fam.log; exit 1; }
  DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
  python3 scripts/rocpd_summary.py $DB bygrid | grep -v "at::\|gated_out" | cut -c1-220 > gpurun_out/${TAG}_exact_route_kernels_$fam.csv
  echo "== $fam"; head -14 gpurun_out/${TAG}_exact_route_kernels_$fam.csv
  rm -rf $OUT
done

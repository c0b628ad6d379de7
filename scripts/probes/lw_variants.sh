# Build-time variants of the wide remainder kernels (dense event lists), timed on one box by their kernel trace:
#   bash scripts/probes/lw_variants.sh "UN OCC" ...   -> gpurun_out/r04_lw_variants.log
[ $# -eq 0 ] && set -- "8 0" "4 0" "4 8" "8 8" "16 0"
export TMPDIR=/tmp
R=$PWD
: > gpurun_out/r04_lw_variants.log
for v in "$@"; do
  set -- $v
  touch modegpt_amd/csrc/cov_i8.hip
  make -C modegpt_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -DMDG_LW_UN=$1 -DMDG_LW_OCC=$2" > /dev/null 2>&1 || { echo "build failed: $v"; exit 1; }
  echo "== events per batch $1, waves per SIMD forced $2" >> gpurun_out/r04_lw_variants.log
  OUT=/tmp/lwv; rm -rf $OUT
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT -o p -- python3 $R/scripts/probes/exact_route_timing.py 14336 32768 silu_gated exact > $R/gpurun_out/r04_lwv.log 2>&1) || { echo "trace failed"; exit 1; }
  DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
  python3 scripts/rocpd_summary.py $DB bygrid | grep "i8_lo_wide" | cut -c20-130 >> gpurun_out/r04_lw_variants.log
  grep -h "whole call" gpurun_out/r04_lwv.log | cut -c1-100 >> gpurun_out/r04_lw_variants.log
done
cat gpurun_out/r04_lw_variants.log

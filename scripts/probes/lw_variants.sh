# Build-time variants of the wide remainder kernels (dense event lists), timed on one box by their kernel trace (the third
# parameter needs scripts/probes/cov_i8_lo_wide_pair_variant.patch applied: two columns per wave, interleaved segment by segment --
# 3.1 + 3.5 ms against 2.6 + 2.8, not shipped):
#   bash scripts/probes/lw_variants.sh "UN OCC PAIR" ...   -> gpurun_out/r04_lw_variants.log
[ $# -eq 0 ] && set -- "8 0 0" "4 0 0" "4 8 0" "8 8 0" "16 0 0" "4 0 1" "8 0 1"
export TMPDIR=/tmp
R=$PWD
: > gpurun_out/r04_lw_variants.log
for v in "$@"; do
  set -- $v
  touch modegpt_amd/csrc/cov_i8.hip
  make -C modegpt_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -DMDG_LW_UN=$1 -DMDG_LW_OCC=$2 -DMDG_LW_PAIR=${3:-0}" > /dev/null 2>&1 || { echo "build failed: $v"; exit 1; }
  echo "== events per batch $1, waves per SIMD forced $2, two columns per wave ${3:-0}" >> gpurun_out/r04_lw_variants.log
  OUT=/tmp/lwv; rm -rf $OUT
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT -o p -- python3 $R/scripts/probes/exact_route_timing.py 14336 32768 silu_gated exact > $R/gpurun_out/r04_lwv.log 2>&1) || { echo "trace failed"; exit 1; }
  DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
  python3 scripts/rocpd_summary.py $DB bygrid | grep "i8_lo_wide" | grep -v gated_out | cut -c20-130 >> gpurun_out/r04_lw_variants.log
  grep -h "whole call" gpurun_out/r04_lwv.log | cut -c1-100 >> gpurun_out/r04_lw_variants.log
done
cat gpurun_out/r04_lw_variants.log

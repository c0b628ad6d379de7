# Build-time variants of the three-plane (exact route) product kernel, timed on one box:
#   bash scripts/probes/p3_variants.sh "KSS RING DEFER" ...     (defaults below)  -> gpurun_out/r04_p3_variants.log
[ $# -eq 0 ] && set -- "2 3 4" "1 3 4" "3 2 4" "2 3 0" "2 3 2" "2 3 6" "2 3 8"
: > gpurun_out/r04_p3_variants.log
for v in "$@"; do
  set -- $v
  touch modegpt_amd/csrc/cov_i8.hip
  make -C modegpt_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -DMDG_I8_KSS3=$1 -DMDG_I8_RING3=$2 -DMDG_I8_DEFER3=$3" > /dev/null 2>&1 || { echo "build failed: $v"; exit 1; }
  echo "== k-steps per stage $1, ring $2, deferred $3" >> gpurun_out/r04_p3_variants.log
  timeout -k 10 200 python3 scripts/probes/exact_route_timing.py 14336 32768 all exact 2>&1 | grep "exact=True" | cut -c1-130 >> gpurun_out/r04_p3_variants.log || exit 1
done
cat gpurun_out/r04_p3_variants.log

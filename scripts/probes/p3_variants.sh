set -x
python scripts/probes/exact_route_timing.py 2>&1 | grep "exact=True" > gpurun_out/r04_p3_ring4_prefetch.log
for v in "3 0" "5 1" "6 1" "4 0"; do
  set -- $v
  touch modegpt_amd/csrc/cov_i8.hip
  make -C modegpt_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -DMDG_I8_RING3=$1 -DMDG_I8_PREFETCH3=$2" > /dev/null 2>&1 || exit 1
  python scripts/probes/exact_route_timing.py 2>&1 | grep "exact=True" > gpurun_out/r04_p3_ring$1_prefetch$2.log
done
tail -n 3 gpurun_out/r04_p3_*.log

# Where the tile kernel and the wide kernels of the exact route's remainder cross over: SiLU-gated columns at the sigma_mlp width,
# calls of 2048 .. 32768 tokens (10 .. 164 listed elements per column), whole call with either implementation forced.
#   bash scripts/probes/lo_mode_crossover.sh   -> gpurun_out/r04_lo_mode_crossover.log
: > gpurun_out/r04_lo_mode_crossover.log
for v in "0 64 wide" "100000 512 tiles"; do
  set -- $v
  touch modegpt_amd/csrc/cov_i8.hip
  make -C modegpt_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -DMDG_LO_SPARSE_MEAN=$1 -DMDG_LO_SPARSE_MAX=$2" > /dev/null 2>&1 || { echo "build failed"; exit 1; }
  for T in 2048 4096 8192 16384 32768; do
    echo "== $3 forced, $T tokens" >> gpurun_out/r04_lo_mode_crossover.log
    timeout -k 10 200 python3 scripts/probes/exact_route_timing.py 14336 $T silu_gated exact 2>&1 | grep "exact=True" | cut -c1-130 >> gpurun_out/r04_lo_mode_crossover.log || exit 1
  done
done
cat gpurun_out/r04_lo_mode_crossover.log

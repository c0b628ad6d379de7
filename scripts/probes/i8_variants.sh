# Build-flag sweep of the int8 product kernel on the GPU box (run through gpurun): rebuilds cov_i8.o with each flag set,
# NOTE (round 3): the -D knobs this script sweeps were moved out of cov_i8.hip; apply scripts/probes/cov_i8_variants.patch first.
# relinks the library and runs scripts/bench_kernels.py covi8 (Gaussian columns: five planes) and covi8p6 (SiLU-gated: six).
#   usage: bash scripts/probes/i8_variants.sh "-DA=1 -DB=2" "-DA=2" ...      (I8_MODE="covi8 covi8p6" by default)
set -e
cd modegpt_amd/csrc
for cfg in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -DMDG_EXPERIMENT $cfg -c cov_i8.hip -o build/cov_i8.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmodegpt_hip.so build/*.o
  echo "== $cfg"
  (cd ../.. && timeout -k 10 200 python3 scripts/bench_kernels.py ${I8_MODE:-covi8 covi8p6} 2>&1 | grep "cov mlp\|cov x\|route\|stamps\|wgtimes\|slowest\|fastest")
done

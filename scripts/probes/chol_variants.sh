# Build-flag sweep of the blocked Cholesky (run through gpurun): rebuilds chol.o with each flag set, relinks, times the MLP chain.
#   usage: bash scripts/probes/chol_variants.sh "-DMDG_CHOL_NBO=1024" ...
set -e
cd modegpt_amd/csrc
for cfg in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w $cfg -c chol.hip -o build/chol.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmodegpt_hip.so build/*.o
  echo "== $cfg"
  (cd ../.. && timeout -k 10 200 python3 scripts/probes/decomp_phases.py phases 2>&1 | grep "potrf_lower\|potrs")
done

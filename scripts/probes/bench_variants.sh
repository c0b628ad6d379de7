# Build-flag sweep of cov_i8.hip measured through bench.py itself (timed loop only, no extra legs) on one GPU box:
#   usage: bash scripts/probes/bench_variants.sh "-DA=1" "-DA=2" ...
set -e
for cfg in "$@"; do
  (cd modegpt_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w $cfg -c cov_i8.hip -o build/cov_i8.o &&
   /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmodegpt_hip.so build/*.o)
  echo "== $cfg"
  timeout -k 10 300 python3 bench.py --no-extra-legs --no-cpu-baseline ${BENCH_ARGS:-} | python3 -c "
import json, sys
o = json.loads(sys.stdin.readline())
print('layers/s %.4f  ms/step %.1f  sigma_mlp launch %.2f ms  frac %.3f' % (o['value'], o['ms_per_step'], o['roofline']['avg_launch_ms'], o['roofline']['frac']))"
done

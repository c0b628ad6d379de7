# Workgroups of the column-maximum pass (token slabs per 128 columns), timed by the kernel trace of one batch of a Llama-3-8B layer:
#   bash scripts/probes/colmax_variants.sh [WGS ...]
[ $# -eq 0 ] && set -- 1 2048 4096 8192 16384
for w in "$@"; do
  touch modegpt_amd/csrc/cov_i8.hip
  make -C modegpt_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -DMDG_COLMAX_WGS=$w" > /dev/null 2>&1 || { echo "build failed"; exit 1; }
  echo "== about $w workgroups"
  bash scripts/probes/trace_batch.sh 2>&1 | grep "colmax\|batch:" | cut -c1-100
done

#!/bin/bash
# rocprofv3 passes over mdg_rope_gather at the Llama-3-8B compressed shape (scripts/bench_kernels.py rope):
#   pass 1: --kernel-trace --stats  -> per-kernel durations      pass 2/3: --pmc FETCH_SIZE / WRITE_SIZE (HBM bytes)
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
for pass in trace fetch write; do
  OUT=$R/gpurun_out/rope_$pass
  rm -rf $OUT
  case $pass in
    trace) ARGS="--kernel-trace --stats" ;;
    fetch) ARGS="--kernel-trace --pmc FETCH_SIZE" ;;
    write) ARGS="--kernel-trace --pmc WRITE_SIZE" ;;
  esac
  timeout -k 10 200 rocprofv3 $ARGS -d $OUT -o p -- python3 scripts/bench_kernels.py rope > $OUT.log 2>&1 || exit 1
  python3 scripts/rocpd_summary.py $(ls $OUT/*.db | head -1) > $R/gpurun_out/rope_$pass.csv || exit 1
done
grep -h "rope_gather" $R/gpurun_out/rope_trace.csv $R/gpurun_out/rope_fetch.csv $R/gpurun_out/rope_write.csv | cut -c1-400

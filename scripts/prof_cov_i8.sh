#!/bin/bash
# rocprofv3 passes over the int8 digit-plane covariance at the Llama-3-8B sigma_mlp shape (scripts/bench_kernels.py covi8: sigma_mlp launches, then sigma_x ones):
#   trace: --kernel-trace --stats     fetch / write: --pmc FETCH_SIZE / WRITE_SIZE (HBM bytes)    mfma: MFMA busy counters
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
for pass in trace fetch write mfma; do
  OUT=$R/gpurun_out/covi8_$pass
  rm -rf $OUT
  case $pass in
    trace) ARGS="--kernel-trace --stats" ;;
    fetch) ARGS="--kernel-trace --pmc FETCH_SIZE" ;;
    write) ARGS="--kernel-trace --pmc WRITE_SIZE" ;;
    mfma)  ARGS="--kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" ;;
  esac
  timeout -k 10 200 rocprofv3 $ARGS -d $OUT -o p -- python3 scripts/bench_kernels.py covi8 > $OUT.log 2>&1 || exit 1
  python3 scripts/rocpd_summary.py $(ls $OUT/*.db | head -1) bygrid > $R/gpurun_out/covi8_$pass.csv || exit 1
  rm -rf $OUT
done
grep -h "i8_\|cov_accum_kernel" $R/gpurun_out/covi8_trace.csv | grep -v gated_out | cut -c1-200
grep -h "i8_syrk" $R/gpurun_out/covi8_fetch.csv $R/gpurun_out/covi8_write.csv $R/gpurun_out/covi8_mfma.csv | grep ran_long | grep "FETCH\|WRITE\|SQ_\|GRBM" | cut -c1-200

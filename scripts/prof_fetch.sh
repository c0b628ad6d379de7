#!/bin/bash
# one rocprofv3 --pmc FETCH_SIZE pass over scripts/bench_kernels.py covi8 with the library as built: L2-miss bytes of the product kernel
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/fetchprof
rm -rf $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o p -- python3 scripts/bench_kernels.py ${1:-covi8} > $OUT.log 2>&1 || exit 1
python3 scripts/rocpd_summary.py $(ls $OUT/*.db | head -1) bygrid | grep "i8_syrk" | grep -v gated_out | cut -c40-200
rm -rf $OUT

#!/bin/bash
# rocprofv3 --kernel-trace --stats over the default bench.py run; summaries land in gpurun_out/ (copy to profiles/).
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/benchprof
rm -rf $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT -o p -- python3 bench.py --no-cpu-baseline > $R/gpurun_out/bench_under_rocprof.log 2>&1 || exit 1
DB=$(ls $OUT/*.db | head -1)
python3 scripts/rocpd_summary.py $DB > $R/gpurun_out/bench_kernel_trace_stats.csv || exit 1
python3 scripts/rocpd_summary.py $DB bygrid > $R/gpurun_out/bench_kernel_trace_bygrid.csv || exit 1
rm -rf $OUT
grep -h "i8_syrk" $R/gpurun_out/bench_kernel_trace_bygrid.csv | cut -c1-160
grep -o '"avg_launch_ms": [0-9.]*' $R/gpurun_out/bench_under_rocprof.log | head -2

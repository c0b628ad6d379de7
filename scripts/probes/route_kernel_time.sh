#!/bin/bash
# per-launch time of the route / clear / column launches of the int8 covariance call (rocprofv3 --kernel-trace of bench_kernels.py covi8)
export TMPDIR=/tmp
R=$PWD
OUT=/tmp/route_prof; rm -rf $OUT
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT -o p -- python3 $R/scripts/bench_kernels.py covi8 > $R/gpurun_out/route_run.log 2>&1) || { tail -5 $R/gpurun_out/route_run.log; exit 1; }
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python3 $R/scripts/rocpd_summary.py $DB bygrid | grep -i "route_kernel\|split_vec\|colmax_vec" | sed -E "s/^\"[^\"]*(route_kernel|split_vec|colmax_vec)[^\"]*\"/\1/"

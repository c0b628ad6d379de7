#!/bin/bash
# rocprofv3 kernel trace of the int8 covariance call (clean and with massive columns) + the whole-call timings; run on the GPU box:
#   bash scripts/probes/prof_cov_i8.sh <tag>      -> gpurun_out/<tag>_kernel_trace.csv / _bygrid.csv / .log
# The program stands directly after `--` (no env / bash -c hop), and the exit code of rocprofv3 is reported: VERDICT r2 item 7.
export TMPDIR=/tmp
R=$PWD
TAG=${1:-r03_cov_i8}
OUT=$R/gpurun_out/${TAG}_prof
rm -rf $OUT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT -o p -- python3 $R/scripts/bench_kernels.py covi8 covi8p6 covi8massive > $R/gpurun_out/${TAG}.log 2>&1
RC=$?
echo "rocprofv3 exit code: $RC" >> $R/gpurun_out/${TAG}.log
cd $R
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python3 scripts/rocpd_summary.py $DB > gpurun_out/${TAG}_kernel_trace.csv
python3 scripts/rocpd_summary.py $DB bygrid > gpurun_out/${TAG}_bygrid.csv
rm -rf $OUT
grep -v "^W2026\|^E2026\|^I2026" gpurun_out/${TAG}.log | tail -15
head -30 gpurun_out/${TAG}_kernel_trace.csv
exit $RC

#!/bin/bash
# rocprofv3 --kernel-trace over the decomposition of one Llama-3-8B layer (scripts/bench_kernels.py cov mlp vo qk): per-kernel,
# per-launch-shape durations -> gpurun_out/decomp_trace_{stats,bygrid}.csv
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/decompprof
rm -rf $OUT
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT -o p -- python3 scripts/bench_kernels.py cov mlp vo qk > $OUT.log 2>&1 || exit 1
DB=$(ls $OUT/*.db | head -1)
python3 scripts/rocpd_summary.py $DB > $R/gpurun_out/decomp_trace_stats.csv
python3 - "$DB" > $R/gpurun_out/decomp_trace_bygrid.csv <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
print("kernel,grid_x,grid_y,grid_z,calls,total_ms,avg_ms")
for r in cur.execute("""select s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e6
    from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id
    where s.kernel_name like '%gemm_f64%' or s.kernel_name like '%potrf%' or s.kernel_name like '%syevj%'
    group by 1,2,3,4 order by 6 desc limit 60"""):
    print('"%s",%d,%d,%d,%d,%.3f,%.4f' % (r[0][:60], r[1], r[2], r[3], r[4], r[5], r[6]))
PY
rm -rf $OUT
head -40 $R/gpurun_out/decomp_trace_bygrid.csv

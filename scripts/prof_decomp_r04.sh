#!/bin/bash
# rocprofv3 --kernel-trace over the decomposition chain ALONE (scripts/probes/decomp_phases.py phases: Cholesky n = 14336, triangular
# inverse, Cholesky r = 10035, substitution, gathered cross term -- each phase by itself on one stream, nothing beside it):
# per-kernel and per-launch-shape durations -> gpurun_out/${1:-r04}_decomposition_{phases.log,kernel_trace_stats.csv,kernel_trace_by_launch_shape.csv}
export TMPDIR=/tmp
TAG=${1:-r04}
R=$PWD
OUT=$R/gpurun_out/decompprof_$TAG
rm -rf $OUT
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT -o p -- python3 scripts/probes/decomp_phases.py phases > $R/gpurun_out/${TAG}_decomposition_phases.log 2>&1 || exit 1
DB=$(ls $OUT/*.db | head -1)
python3 scripts/rocpd_summary.py $DB > $R/gpurun_out/${TAG}_decomposition_kernel_trace_stats.csv
python3 - "$DB" > $R/gpurun_out/${TAG}_decomposition_kernel_trace_by_launch_shape.csv <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
print("kernel,grid_x,grid_y,grid_z,calls,total_ms,avg_ms,min_ms,max_ms")
for r in cur.execute("""select s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e6,
    min(d.end-d.start)/1e6, max(d.end-d.start)/1e6
    from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id
    where s.kernel_name like '%gemm_f64%' or s.kernel_name like '%potrf%' or s.kernel_name like '%syevj%' or s.kernel_name like '%lower_%' or s.kernel_name like '%place_inv%'
    group by 1,2,3,4 order by 6 desc limit 80"""):
    print('"%s",%d,%d,%d,%d,%.3f,%.4f,%.4f,%.4f' % (r[0][:70], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8]))
PY
rm -rf $OUT
grep -v "^W2026\|amdgpu.ids" $R/gpurun_out/${TAG}_decomposition_phases.log | tail -12
head -8 $R/gpurun_out/${TAG}_decomposition_kernel_trace_stats.csv | cut -c1-160

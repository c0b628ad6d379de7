#!/bin/bash
# PMC pass over the fused covariance launch (sigma_mlp + sigma_x + sigma_q + sigma_k of one batch): rocprofv3 --pmc <counters> on scripts/bench_kernels.py cov
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/$1; shift
rm -rf $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" -d $OUT -o p -- python3 scripts/bench_kernels.py cov > $OUT.log 2>&1
python3 - <<PY
import sqlite3,glob
db=sqlite3.connect(glob.glob("$OUT/*.db")[0]); cur=db.cursor()
q="""select d.id, (d.end-d.start)/1e6, p.name, sum(e.value) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id=p.id
join rocpd_kernel_dispatch d on e.event_id=d.event_id join rocpd_info_kernel_symbol s on d.kernel_id=s.id
where s.kernel_name like '%cov_accum_multi%' group by d.id, p.name order by d.id, p.name"""
rows=list(cur.execute(q))
first=rows[0][0]
for r in rows:
    if r[0]==first: print("%-32s %.6g   (dispatch %.2f ms)"%(r[2], r[3], r[1]))
PY

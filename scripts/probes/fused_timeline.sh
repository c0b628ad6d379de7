export TMPDIR=/tmp
R=$PWD
OUT=/tmp/fused_prof; rm -rf $OUT
cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT -o p -- python3 $R/scripts/bench_kernels.py covi8fused > $R/gpurun_out/r03_fused.log 2>&1
cd $R
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python3 - $DB <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select s.kernel_name, d.start, d.end, d.grid_size_x from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"))
# find the last occurrence of a fused multi call: sequence starting at a colmax with 3 statistics... print the last 60 dispatches with gaps
last = rows[-70:]
t0 = last[0][1]
prev_end = None
for name, st, en, g in last:
    short = name.split("N_1")[-1][:34] if "N_1" in name else name[:34]
    gap = (st - prev_end) / 1e3 if prev_end else 0
    print(f"{(st - t0)/1e3:9.1f} us  +gap {gap:6.1f}  dur {(en-st)/1e3:8.1f} us  grid {g:7d}  {short}")
    prev_end = en
PY
grep "fused" gpurun_out/r03_fused.log | tail -2

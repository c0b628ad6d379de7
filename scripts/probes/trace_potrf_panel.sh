export TMPDIR=/tmp
R=$PWD
OUT=/tmp/paneltrace; rm -rf $OUT
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT -o p -- python3 $R/scripts/probes/decomp_phases.py potrfonly > $R/gpurun_out/r04_panel_trace.log 2>&1
cd $R
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python3 - $DB <<PY
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("""select s.kernel_name, d.grid_size_x, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"""))
rows = [r for r in rows if 'potrf' in r[0] or 'gemm_f64' in r[0]]
gemms = [i for i, r in enumerate(rows) if 'gemm_f64' in r[0]]
# the second outer panel of the last factorisation: launches between the 6th-last and 5th-last GEMM
a, b = gemms[-6], gemms[-5]
prev_end = rows[a][3]
for r in rows[a + 1:b + 1]:
    nm = 'solve ' if 'ILi0E' in r[0] else 'update' if 'ILi1E' in r[0] else 'diag  ' if 'diag' in r[0] else 'GEMM  '
    print(f"{nm} wgs {r[1]//(1024 if 'potrf' in r[0] else 256):5d}  {(r[3]-r[2])/1e3:8.1f} us   gap before {(r[2]-prev_end)/1e3:6.1f} us")
    prev_end = r[3]
PY

# The launches of one phase of the decomposition in stream order, with their durations (rocprofv3 kernel trace of the LAST call):
#   bash scripts/probes/trace_inverse.sh [inverseonly|potrsonly] [launches to list]
export TMPDIR=/tmp
R=$PWD
OUT=/tmp/invtrace; rm -rf $OUT
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT -o p -- python3 $R/scripts/probes/decomp_phases.py ${1:-inverseonly} > $R/gpurun_out/r04_inv_trace.log 2>&1
cd $R
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python3 - $DB <<PY
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("""select s.kernel_name, d.grid_size_x, d.grid_size_y, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"""))
# the last call of the inverse: take the last 40 mdg kernels
rows = [r for r in rows if 'mdg' in r[0]]
for r in rows[-int("${2:-34}"):]:
    print(f"{r[0][8:60]:52s} grid {r[1]//256:6d} x {r[2]:3d}  {(r[4]-r[3])/1e6:8.3f} ms")
PY

# The blocked Cholesky at n = 14336, launch by launch: per outer panel the time of its inner steps (solve + update launches, their gaps
# included) and of its trailing GEMM, from a rocprofv3 kernel trace of the LAST factorisation.   bash scripts/probes/trace_potrf.sh
export TMPDIR=/tmp
R=$PWD
OUT=/tmp/potrftrace; rm -rf $OUT
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT -o p -- python3 $R/scripts/probes/decomp_phases.py potrfonly > $R/gpurun_out/r04_potrf_trace.log 2>&1
cd $R
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python3 - $DB <<PY
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("""select s.kernel_name, d.grid_size_x, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"""))
rows = [r for r in rows if 'potrf' in r[0] or 'gemm_f64' in r[0]]
# the last factorisation: walk back from the end to its first launch (a potrf_diag launch with nothing before it in the call)
gemms = [i for i, r in enumerate(rows) if 'gemm_f64' in r[0]]
last = rows[gemms[-7] - 40:] if len(gemms) >= 7 else rows
# find the call start: the 7th GEMM from the end is the first panel's; its inner steps precede it
start = gemms[-7]
while start > 0 and 'gemm_f64' not in rows[start - 1][0]:
    start -= 1
seq = rows[start:]
t0 = seq[0][2]
panel, inner_t0, n_inner, busy = 0, seq[0][2], 0, 0
for r in seq:
    if 'gemm_f64' in r[0]:
        print(f"panel {panel}: {n_inner:3d} inner launches {(r[2]-inner_t0)/1e6:7.3f} ms wall ({busy/1e6:7.3f} ms inside kernels)   trailing GEMM {r[1]//256:6d} tiles {(r[3]-r[2])/1e6:7.3f} ms")
        panel += 1; inner_t0 = r[3]; n_inner = 0; busy = 0
    else:
        n_inner += 1; busy += r[3] - r[2]
print(f"last panel: {n_inner} inner launches {(seq[-1][3]-inner_t0)/1e6:7.3f} ms wall ({busy/1e6:7.3f} ms inside kernels);  whole factorisation {(seq[-1][3]-t0)/1e6:7.3f} ms")
PY

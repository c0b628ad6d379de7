# One calibration batch of a Llama-3-8B layer through ops.cov_accum_multi (sigma_mlp in its launch, sigma_x / q / k in theirs),
# launch by launch in stream order with durations and gaps (rocprofv3 kernel trace of the LAST batch).   bash scripts/probes/trace_batch.sh
export TMPDIR=/tmp
R=$PWD
OUT=/tmp/batchtrace; rm -rf $OUT
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT -o p -- python3 $R/scripts/probes/fused_all_four.py gaussian > $R/gpurun_out/r04_batch_trace.log 2>&1
cd $R
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python3 - $DB <<PY
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("""select s.kernel_name, d.grid_size_x, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"""))
rows = [r for r in rows if 'mdg' in r[0] or 'rocclr' in r[0]]
# the split variant runs third (warm + timed): find the last sigma_mlp-sized i8_syrk<3> launch that is followed by another <3> launch (the small one)
big = [i for i, r in enumerate(rows) if 'i8_syrk_kernelILi3' in r[0] and (r[3] - r[2]) > 10e6]
# batches of the 'split' variant: a big launch followed by a small <3> launch before the next big one
def batch_of(i):
    j = i
    while j > 0 and 'colmax' not in rows[j][0]: j -= 1
    while j > 0 and ('colmax' in rows[j - 1][0] or 'fillBuffer' in rows[j - 1][0] or 'memset' in rows[j-1][0].lower()): j -= 1
    return j
for idx in reversed(big):
    nxt = [k for k in range(idx + 1, min(idx + 80, len(rows))) if 'i8_syrk_kernelILi3' in rows[k][0] and (rows[k][3] - rows[k][2]) > 0.5e6]
    if nxt and (rows[nxt[0]][3] - rows[nxt[0]][2]) < 5e6:
        a = batch_of(idx)
        b = nxt[0]
        while b + 1 < len(rows) and 'colmax' not in rows[b + 1][0]: b += 1
        prev = rows[a][2]
        tot = 0
        for r in rows[a:b + 1]:
            d = (r[3] - r[2]) / 1e3
            print(f"{r[0][8:58]:50s} {d:9.1f} us   gap {(r[2]-prev)/1e3:7.1f} us")
            prev = r[3]; tot += d
        print(f"batch: {(rows[b][3]-rows[a][2])/1e6:.3f} ms wall, {tot/1e3:.3f} ms inside kernels")
        break
PY

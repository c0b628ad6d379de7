"""Timeline of one blocked Cholesky (n = 14336) from a rocprofv3 --kernel-trace database: per inner step the kernels' start / end
relative to the step's diagonal-block kernel, by queue.  usage: python3 potrf_timeline.py <rocpd .db> [first_step n_steps]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(cur.execute(f"""select d.start, d.end, s.kernel_name, d.grid_size_x, {qcol or 0} from rocpd_kernel_dispatch d
    join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"""))
# the LAST run of >= 100 potrf_diag kernels = the factorisation to look at
diag = [i for i, r in enumerate(rows) if "potrf_diag" in r[2]]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 40
count = int(sys.argv[3]) if len(sys.argv) > 3 else 4
# take the final factorisation of size 112 blocks: find the last index where 112 consecutive diag kernels precede
end_i = diag[-1]
start_i = diag[-112] if len(diag) >= 112 else diag[0]
t0 = rows[start_i][0]
print("whole factorisation: %.3f ms, %d kernels" % ((rows[end_i][1] - t0) / 1e6, end_i - start_i + 1))
d_idx = [i for i in diag if i >= start_i]
for k in range(first, first + count):
    a, b = d_idx[k], d_idx[k + 1]
    base = rows[a][0]
    print("step %d (starts %.3f ms):" % (k, (base - t0) / 1e6))
    for r in rows[a:b]:
        print("   q%-3s %8.1f .. %8.1f us  %-28s grid %d" % (r[4], (r[0] - base) / 1e3, (r[1] - base) / 1e3, r[2].split("(")[0][-40:].replace("_ZN3mdg", "")[:28], r[3]))
print("kernels longer than 250 us (start .. end in ms from the first diagonal block), and the start of every 8th step:")
for r in rows[start_i:end_i + 1]:
    if r[1] - r[0] > 250e3:
        print("   q%-3s %8.3f .. %8.3f ms  (%.0f us) grid %d" % (r[4], (r[0] - t0) / 1e6, (r[1] - t0) / 1e6, (r[1] - r[0]) / 1e3, r[3]))
print("step starts (ms):", " ".join("%d:%.2f" % (k, (rows[d_idx[k]][0] - t0) / 1e6) for k in range(0, len(d_idx), 8)))

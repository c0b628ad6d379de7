"""Summarise a rocprofv3 rocpd .db: per-kernel dispatch stats (and PMC counter sums if present) as CSV on stdout.
With a second argument `bygrid`, kernels are additionally split by their grid size (one row per launch shape) and by whether
the launch did work: mdg_cov_accum_i8 enqueues the five-plane product, the six-plane product and the fp64 kernel for every
call and the device picks one -- the other launches exit on their first instruction (a few microseconds); the same holds for the
exact route's remainder kernels (tile kernel for sparse event lists, wide kernels + x_d copy for dense ones).  Dispatches shorter
than GATE_MS are listed as `gated_out` rows so that the averages of the launches that ran are not diluted.  Since the persistent
launch gives every product launch the same grid (256 workgroups), the i8_syrk launches that ran are further split by duration:
`ran_long` (>= LONG_MS: the sigma_mlp-sized statistics) and `ran` (sigma_x-sized)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
BYGRID = len(sys.argv) > 2 and sys.argv[2] == "bygrid"
GATE_MS = 0.03
LONG_MS = 10.0
GATED = ("(s.kernel_name like '%%i8_syrk_kernel%%' or s.kernel_name like '%%cov_accum_kernel%%' or s.kernel_name like '%%i8_lo_%%' or "
         "s.kernel_name like '%%i8_copy_xd%%' or s.kernel_name like '%%i8_patch_xd%%' or s.kernel_name like '%%i8_residue_lo%%') and (d.end-d.start) < %d"
         % int(GATE_MS * 1e6))
LONG = "s.kernel_name like '%%i8_syrk_kernel%%' and (d.end-d.start) >= %d" % int(LONG_MS * 1e6)
CLASS = "case when %s then 'gated_out' when %s then 'ran_long' else 'ran' end" % (GATED, LONG)
if BYGRID:
    print("kernel,grid_x,class,calls,avg_ms,min_ms,max_ms")
    for r in cur.execute("""select s.kernel_name, d.grid_size_x, %s, count(*),
            avg(d.end-d.start)/1e6, min(d.end-d.start)/1e6, max(d.end-d.start)/1e6
            from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id
            group by 1, 2, 3 order by 5 desc""" % CLASS):
        print('"%s",%d,%s,%d,%.4f,%.4f,%.4f' % r)
    try:
        rows = list(cur.execute("""select s.kernel_name, d.grid_size_x, %s, p.name,
            count(distinct d.id), sum(e.value) / count(distinct d.id)
            from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id=p.id join rocpd_kernel_dispatch d on e.event_id=d.event_id
            join rocpd_info_kernel_symbol s on d.kernel_id=s.id group by 1, 2, 3, 4 order by 1, 2, 3, 4""" % CLASS))
        if rows:
            print("\nkernel,grid_x,class,counter,dispatches,sum_per_dispatch")
            for r in rows:
                print('"%s",%d,%s,%s,%d,%.6g' % r)
    except sqlite3.Error as e:
        print("# no pmc tables:", e)
    sys.exit(0)
tot = cur.execute("select sum(end-start) from rocpd_kernel_dispatch").fetchone()[0]
print("kernel,calls,total_ms,avg_ms,min_ms,max_ms,pct")
q = """select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e6, min(d.end-d.start)/1e6,
       max(d.end-d.start)/1e6 from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id
       group by s.kernel_name order by 3 desc"""
for r in cur.execute(q):
    print('"%s",%d,%.3f,%.4f,%.4f,%.4f,%.2f' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] * 1e6 / tot))
try:
    rows = list(cur.execute("""select s.kernel_name, p.name, count(*), sum(e.value), avg(e.value)
        from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id=p.id
        join rocpd_kernel_dispatch d on e.event_id=d.event_id
        join rocpd_info_kernel_symbol s on d.kernel_id=s.id group by s.kernel_name, p.name order by 1, 2"""))
    if rows:
        print("\nkernel,counter,dispatches,sum,avg_per_dispatch")
        for r in rows:
            print('"%s",%s,%d,%.6g,%.6g' % r)
except sqlite3.Error as e:
    print("# no pmc tables:", e)

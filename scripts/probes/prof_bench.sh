#!/bin/bash
# The judged profile: rocprofv3 --kernel-trace --stats of the DEFAULT bench command, then three separate PMC passes over the int8
# covariance call at the bench's sigma_mlp shape (counters never ride along a timed run; --pmc only with --kernel-trace).
#   bash scripts/probes/prof_bench.sh <tag> [bench args]     -> gpurun_out/<tag>_bench_kernel_trace_stats.csv, _by_launch_shape.csv,
#                                                               _bench_under_rocprof.log, <tag>_cov_i8_pmc.csv, <tag>_cov_i8_hbm_traffic.json
# Copy what should be judged into profiles/.  The program stands directly after `--`.
export TMPDIR=/tmp
R=$PWD
TAG=${1:-r03}
shift
OUT=/tmp/${TAG}_prof
rm -rf $OUT
cd /tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $OUT -o p -- python3 $R/bench.py "$@" > $R/gpurun_out/${TAG}_bench_under_rocprof.log 2>&1
echo "rocprofv3 exit code: $?" >> $R/gpurun_out/${TAG}_bench_under_rocprof.log
cd $R
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python3 scripts/rocpd_summary.py $DB > gpurun_out/${TAG}_bench_kernel_trace_stats.csv
python3 scripts/rocpd_summary.py $DB bygrid > gpurun_out/${TAG}_bench_kernel_trace_by_launch_shape.csv
rm -rf $OUT
grep -h "i8_syrk_kernelILi\|i8_lo_" gpurun_out/${TAG}_bench_kernel_trace_by_launch_shape.csv | grep -v gated_out | cut -c1-200
: > gpurun_out/${TAG}_cov_i8_pmc.csv
for pass in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"; do
  rm -rf $OUT
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass -d $OUT -o p -- python3 $R/scripts/bench_kernels.py covi8 covi8p6 > $R/gpurun_out/${TAG}_pmc_pass.log 2>&1) || { echo "PMC pass $pass failed"; tail -5 gpurun_out/${TAG}_pmc_pass.log; exit 1; }
  DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
  echo "# pass: $pass" >> gpurun_out/${TAG}_cov_i8_pmc.csv
  python3 scripts/rocpd_summary.py $DB bygrid | grep -i "i8_syrk\|i8_lo_\|i8_copy_xd\|^kernel" >> gpurun_out/${TAG}_cov_i8_pmc.csv
  rm -rf $OUT
done
python3 - <<PY
import csv, json, re
rows = [l.strip() for l in open("gpurun_out/${TAG}_cov_i8_pmc.csv")]
val = {}
for l in rows:
    m = re.match(r'"(.*i8_syrk_kernelILi(\d)EE.*)",(\d+),(ran_long),(\w+),(\d+),([0-9.e+]+)', l)
    if m:
        val[(m.group(2), m.group(5))] = float(m.group(7))
    m = re.match(r'"(.*i8_syrk_kernelILi(\d)EE.*)",(\d+),(ran_long),(\d+),([0-9.]+),', l)
    if m:
        val[(m.group(2), "avg_ms")] = float(m.group(6))
# the remainder kernels of the exact route (class "ran": not i8_syrk launches), each at its sigma_mlp-sized grid: the tile kernel ran on
# the Gaussian call (sparse lists), the two wide kernels and the x_d copy on the SiLU-gated one (dense lists)
for key, pat in (("lo", "i8_lo_product_kernel"), ("wide0", "i8_lo_wide_kernelILb0"), ("wide1", "i8_lo_wide_kernelILb1"), ("copy", "i8_copy_xd")):
    grid = max([int(m.group(2)) for m in (re.match(r'"(.*%s.*)",(\d+),(ran),' % pat, l) for l in rows) if m] or [0])
    for l in rows:
        m = re.match(r'"(.*%s.*)",%d,(ran),(\w+),(\d+),([0-9.e+]+)' % (pat, grid), l)
        if m and not m.group(3).isdigit():
            val[(key, m.group(3))] = float(m.group(5))
        m = re.match(r'"(.*%s.*)",%d,(ran),(\d+),([0-9.]+),' % (pat, grid), l)
        if m:
            val[(key, "avg_ms")] = float(m.group(4))
out = {}
for key, name, note in (("lo", "remainder_tile_kernel", "i8_lo_product_kernel (sparse lists: the Gaussian call)"),
                        ("wide0", "remainder_wide_kernel_rows", "i8_lo_wide_kernel<false> (dense lists: the SiLU-gated call)"),
                        ("wide1", "remainder_wide_kernel_columns", "i8_lo_wide_kernel<true>"), ("copy", "x_d_copy", "i8_copy_xd_kernel")):
    if (key, "FETCH_SIZE") in val:
        out[name] = {"fetch_bytes_corrected": val[(key, "FETCH_SIZE")] * 1024 * 2, "write_bytes": val.get((key, "WRITE_SIZE"), 0.0) * 1024,
                     "avg_ms_under_profiler": val.get((key, "avg_ms")), "note": note + ", launches that did the work"}
for P in ("3", "5", "6"):
    if (P, "FETCH_SIZE") in val:
        fetch, write = val[(P, "FETCH_SIZE")] * 1024 * 2, val.get((P, "WRITE_SIZE"), 0.0) * 1024
        out["planes_" + P] = {"fetch_bytes_corrected": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
                              "avg_ms_under_profiler": val.get((P, "avg_ms")),
                              "mfma_busy": val.get((P, "SQ_VALU_MFMA_BUSY_CYCLES"), 0) / max(val.get((P, "GRBM_GUI_ACTIVE"), 1) * 128, 1),
                              "wait_any_share": val.get((P, "SQ_WAIT_ANY"), 0) / max(val.get((P, "SQ_WAVE_CYCLES"), 1), 1),
                              "lds_bank_conflict": val.get((P, "SQ_LDS_BANK_CONFLICT")),
                              "clock_ghz": (val.get((P, "GRBM_GUI_ACTIVE"), 0) / 8 / (val.get((P, "avg_ms"), 1) * 1e-3) / 1e9) if val.get((P, "avg_ms")) else None}
res = {"source": "scripts/probes/prof_bench.sh: rocprofv3 --kernel-trace --pmc <one group per pass> -- python3 scripts/bench_kernels.py covi8 covi8p6; "
                 "sigma_mlp-sized dispatches (class ran_long) of i8_syrk_kernel<3> (the exact route's nine-pair launch: Gaussian and SiLU-gated columns both take it at "
                 "this width; <5> / <6> appear when the truncated product ran) averaged; raw rows: the _cov_i8_pmc.csv beside this file",
       "units": "bytes per launch; FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B), WRITE_SIZE (KB) as is",
       "hbm_bytes_per_launch": (out.get("planes_3") or out.get("planes_5") or {}).get("hbm_bytes_per_launch"), **out}
json.dump(res, open("gpurun_out/${TAG}_cov_i8_hbm_traffic.json", "w"), indent=1)
print(json.dumps(res)[:1500])
PY

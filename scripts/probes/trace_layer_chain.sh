# One layer's whole decomposition chain (compress_nystrom + compress_qk + compress_vo through engine.compress_layer) on one stream:
# kernels by total time, from a rocprofv3 kernel trace of scripts/probes/decomp_phases.py async.   bash scripts/probes/trace_layer_chain.sh
export TMPDIR=/tmp
R=$PWD
OUT=/tmp/chaintrace; rm -rf $OUT
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT -o p -- python3 $R/scripts/probes/decomp_phases.py async > $R/gpurun_out/r04_chain_trace.log 2>&1
cd $R
grep "layer chain" gpurun_out/r04_chain_trace.log
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python3 - $DB <<PY
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("""select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"""))
# the chains: engine.compress_layer runs three times (warm-up, deferred, checked); take the middle third of the mdg kernels after the covariance
names = [r for r in rows if 'mdg' in r[0] and 'i8_' not in r[0] and 'cov_' not in r[0]]
third = len(names) // 3
seq = names[third:2 * third]
tot = {}
for n_, s_, e_ in seq:
    k = n_[4:70]
    tot[k] = tot.get(k, [0, 0.0]); tot[k][0] += 1; tot[k][1] += (e_ - s_) / 1e6
print(f"one chain: {len(seq)} launches, {(seq[-1][2]-seq[0][1])/1e6:.1f} ms wall, {sum(v[1] for v in tot.values()):.1f} ms inside kernels")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{k:66s} {v[0]:5d} launches {v[1]:8.2f} ms")
PY

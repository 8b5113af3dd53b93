#!/bin/bash
# Kernel trace of a bench workload (inference): usage  WL=vr_fusion bash tools/debug/wl_prof.sh
R=$(pwd); OUT=$R/gpurun_out/prof_wl; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o stats -- python $R/bench.py --workload ${WL:-vr_fusion} --steps 10 --warmup 3 --no-cpu-baseline --no-extra-passes --no-kernel-timing > $OUT/bench.log 2>&1
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$OUT/trace/stats_kernel_trace.csv")))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last 40 % of the launches are steady-state steps
rows = rows[int(len(rows) * 0.6):]
span = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e6
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    agg[r['Kernel_Name'][:110]][0] += d
    agg[r['Kernel_Name'][:110]][1] += 1
tot = sum(v[0] for v in agg.values()) / 1e6
print("window %.1f ms, kernel time %.1f ms (%.0f %% busy), %d launches" % (span, tot, 100 * tot / span, len(rows)))
for k, (d, n) in sorted(agg.items(), key=lambda x: -x[1][0])[:40]:
    print("%6.2f %%  %9.1f us %5d  %s" % (100 * d / 1e6 / tot, d / 1e3, n, k))
PY
tail -1 $OUT/bench.log | cut -c1-300

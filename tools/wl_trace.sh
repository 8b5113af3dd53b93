#!/bin/bash
# kernel trace of a side workload: tools/wl_trace.sh <workload> -> gpurun_out/wl_<workload>/{stats.csv,timeline.txt}
R=$(pwd); OUT=$R/gpurun_out/wl_$1; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o stats -- python $R/bench.py --workload $1 --steps 6 --warmup 3 --no-cpu-baseline --no-extra-passes --no-side-configs --no-kernel-timing > $OUT/bench.log 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/stats.csv
python - $(find $OUT/trace -name "*kernel_trace.csv" | head -1) $2 > $OUT/timeline.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2] if len(sys.argv) > 2 else "vox_insert"
marks = [int(r["Start_Timestamp"]) for r in rows if key in r["Kernel_Name"]]
t_mid = marks[-2] if len(marks) > 1 else int(rows[len(rows) // 2]["Start_Timestamp"])
seg = [r for r in rows if t_mid - 500_000 < int(r["Start_Timestamp"]) < t_mid + 9_000_000]
t0 = int(seg[0]["Start_Timestamp"]); q = {}
for r in seg:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if en - st < 15000: continue
    print("%9.1f %8.1f q%d %s" % ((st - t0) / 1e3, (en - st) / 1e3, q.setdefault(r.get("Queue_Id"), len(q)), r["Kernel_Name"].replace("void ", "").replace("df3d::", "")[:100]))
PY
rm -rf $OUT/trace
tail -1 $OUT/bench.log | cut -c1-200

#!/bin/bash
# rocprofv3 kernel statistics of tools/debug/msda_bwd_probe.py (13 backward calls)
R=$(pwd); OUT=$R/gpurun_out/msda_bwd_trace; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python $R/tools/debug/msda_bwd_probe.py > $OUT/log 2>&1
tail -1 $OUT/log
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print("calls %5d  avg %8.1f us  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY

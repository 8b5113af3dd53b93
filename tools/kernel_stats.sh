#!/bin/bash
# rocprofv3 kernel statistics of the headline bench (short): prints the rows of the kernels matching $1 (regex)
R=$(pwd); OUT=$R/gpurun_out/kstats_$$; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python $R/bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-extra-passes --no-side-configs --no-kernel-timing > $OUT/log 2>&1
grep -h "^{" $OUT/log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('ms_per_step', d['ms_per_step'])"
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" "$1" <<'PY'
import csv, re, sys
pat = re.compile(sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    if pat.search(r["Name"]):
        print("%-60s calls %4s avg %8.1f min %8.1f max %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
rm -rf $OUT

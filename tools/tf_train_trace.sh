#!/bin/bash
# rocprofv3 kernel statistics of the TransFusion training step (bench.py --workload tf_fusion --stage train)
R=$(pwd); OUT=$R/gpurun_out/tf_train_trace; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
STEPS=${STEPS:-4}; WARM=${WARM:-2}; FR=${FR:-2}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python $R/bench.py --workload tf_fusion --stage train --steps $STEPS --warmup $WARM --frames $FR --no-cpu-baseline --no-extra-passes --no-side-configs --no-kernel-timing $EXTRA > $OUT/log 2>&1
grep -h "^{" $OUT/log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('ms_per_step', d['ms_per_step'])"
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
N=$((STEPS+WARM+FR)) python - "$f" <<'PY'
import csv, os, sys
n = float(os.environ["N"])
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per step (%d steps incl. setup / warm-up): %.2f ms; launches per step %.0f" % (n, tot / n / 1e6, sum(int(r["Calls"]) for r in rows) / n))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:60]:
    print("%6.2f ms/step  calls/step %6.1f  avg %8.1f us  %s" % (float(r["TotalDurationNs"]) / n / 1e6, int(r["Calls"]) / n, float(r["AverageNs"]) / 1e3, r["Name"][:120]))
PY

#!/bin/bash
# rocprofv3 kernel statistics of the training step (bench.py --stage train): top kernels by total time, grouped
R=$(pwd); OUT=$R/gpurun_out/train_trace; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python $R/bench.py --workload ${WL:-centerpoint} --stage train --steps 4 --warmup 2 --no-cpu-baseline --no-extra-passes --no-side-configs --no-kernel-timing > $OUT/log 2>&1
grep -h "^{" $OUT/log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('ms_per_step', d['ms_per_step'])"
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per step (6 steps incl. warmup): %.2f ms" % (tot / 6e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    print("%6.2f ms/step  calls/step %6.1f  avg %8.1f us  %s" % (float(r["TotalDurationNs"]) / 6e6, int(r["Calls"]) / 6.0, float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY

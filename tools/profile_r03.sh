#!/bin/bash
# rocprofv3 kernel traces behind profiles/r03_* (run on an MI355X box from the repo root; writes gpurun_out/prof_r03/).
# usage: tools/profile_r03.sh [tag] [extra bench.py args...]     e.g.  tools/profile_r03.sh hot --stage hot_path
set -e
R=$(pwd)
TAG=${1:-detect}
shift || true
OUT=$R/gpurun_out/prof_r03/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-passes --no-kernel-timing $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o stats -- $BENCH > $OUT/bench_trace.log 2>&1
cd $R
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_summary.py $T > $OUT/summary.txt
python tools/trace_timeline.py $T > $OUT/timeline.txt
tail -3 $OUT/bench_trace.log
head -50 $OUT/summary.txt

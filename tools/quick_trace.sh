#!/bin/bash
# one rocprofv3 kernel-trace pass of the detector step + its timeline (gpurun_out/<tag>/timeline.txt, stats.csv)
R=$(pwd); OUT=$R/gpurun_out/${1:-qtrace}; mkdir -p $OUT; shift
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o stats -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra-passes --no-side-configs --no-kernel-timing "$@" > $OUT/bench.log 2>&1
cd $R
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_timeline.py $T > $OUT/timeline.txt
S=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cp $S $OUT/stats.csv
rm -rf $OUT/trace
tail -1 $OUT/bench.log | cut -c1-300

set -e
R=$(pwd)
OUT=$R/gpurun_out/prof_train
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o stats -- python $R/bench.py --workload ${WL:-cp_lidar} --stage train --steps 10 --warmup 3 --no-cpu-baseline --no-extra-passes --no-kernel-timing > $OUT/bench.log 2>&1
find $OUT -name "*kernel_stats.csv" | head
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
python $R/tools/debug/step_kernels.py $OUT/trace/stats_kernel_trace.csv 45

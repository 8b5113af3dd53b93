#!/bin/bash
# The rocprofv3 passes behind profiles/r03_* (run on an MI355X box from the repo root; writes gpurun_out/prof_r03pmc/).
# Counters are collected in their own passes with --kernel-trace only (no sys/runtime/hip tracing next to --pmc).
set -e
R=$(pwd)
OUT=$R/gpurun_out/prof_r03pmc
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-passes"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o stats -- $BENCH --no-kernel-timing > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- $BENCH --no-kernel-timing > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- $BENCH --no-kernel-timing > $OUT/bench_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib_fetch -o fetch -- python $R/tools/ubench/pmc_calib.py > $OUT/calib_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/calib_write -o write -- python $R/tools/ubench/pmc_calib.py > $OUT/calib_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -o sq -- $BENCH --no-kernel-timing > $OUT/bench_sq.log 2>&1 || true
cd $R
python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err || true
find $OUT -name "*.csv" | head -30
# A / B of the round-3 change "layers whose fp32 rows nobody reads write split rows only": the same WRITE_SIZE pass with every
# layer writing both formats again (DF3D_EXEC_F32_ALL=1, round 2's behaviour)
cd /tmp
DF3D_EXEC_F32_ALL=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_all -o write -- $BENCH --no-kernel-timing > $OUT/bench_write_all.log 2>&1
cd $R

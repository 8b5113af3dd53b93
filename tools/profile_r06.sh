#!/bin/bash
# The rocprofv3 passes behind profiles/r06_* (run on an MI355X box from the repo root; writes gpurun_out/prof_r06/).
# Counters are collected in their own passes with --kernel-trace only (no sys/runtime/hip tracing next to --pmc).
# usage: tools/profile_r06.sh [tag]   (tag = sub-directory under gpurun_out/, default prof_r06)
set -e
R=$(pwd)
OUT=$R/gpurun_out/${1:-prof_r06}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-passes --no-side-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o stats -- $BENCH --no-kernel-timing > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- $BENCH --no-kernel-timing > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- $BENCH --no-kernel-timing > $OUT/bench_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib_fetch -o fetch -- python $R/tools/ubench/pmc_calib.py > $OUT/calib_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/calib_write -o write -- python $R/tools/ubench/pmc_calib.py > $OUT/calib_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -o sq -- $BENCH --no-kernel-timing > $OUT/bench_sq.log 2>&1 || true
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/sq2 -o sq2 -- $BENCH --no-kernel-timing > $OUT/bench_sq2.log 2>&1 || true
cd $R
python bench.py --no-cpu-baseline --no-side-configs > $OUT/bench.json 2> $OUT/bench.err || true
# keep what is merged back small: the counter CSVs of the conv / rulebook / fusion kernels only
for d in fetch write sq sq2 calib_fetch calib_write; do
  f=$(find $OUT/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
keep = [r for r in rows if "df3d" in r["Kernel_Name"]]
cols = ["Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value"]
cols = [c for c in cols if rows and c in rows[0]]
w = csv.DictWriter(open(f, "w"), cols)
w.writeheader()
for r in keep:
    w.writerow({c: r[c] for c in cols})
PY
done
find $OUT -name "*.csv" | head -40

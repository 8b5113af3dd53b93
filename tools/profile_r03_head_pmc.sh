#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes with --kernel-trace only) of the head's final-convolution kernels on
# tools/ubench/head_final.py: the round-3 matrix-core kernel and the round-2 vector-ALU kernel.  Run on an MI355X box from the
# repo root; writes gpurun_out/prof_r03head/.
set -e
R=$(pwd)
OUT=$R/gpurun_out/prof_r03head
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for mode in mfma valu; do
  for c in FETCH_SIZE WRITE_SIZE; do
    DF3D_HEADFINAL=$mode rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${mode}_$c -o pmc -- python $R/tools/ubench/head_final.py > $OUT/${mode}_$c.log 2>&1
  done
done
cd $R
find $OUT -name "*counter_collection.csv" | head

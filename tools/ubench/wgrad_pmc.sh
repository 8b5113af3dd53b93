#!/bin/bash
# Counter passes over the filter-gradient kernels on conv4 / conv3 and the linear-layer shapes (tools/ubench/wgrad3_probe.py):
# LDS bank conflicts of the transposing reads, wave-state cycles, MFMA busy.  Counters only next to --kernel-trace.
# usage: wgrad_pmc.sh <tag>
R=$(pwd); OUT=$R/gpurun_out/wgrad_pmc_$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o pmc -- python $R/tools/ubench/wgrad3_probe.py > $OUT/g$i.log 2>&1 || echo "group $i failed"
done
cd $R
python tools/ubench/pmc_avg.py $OUT --match wgrad > $OUT/summary.txt 2>&1
cat $OUT/summary.txt

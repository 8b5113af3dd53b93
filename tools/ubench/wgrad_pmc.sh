#!/bin/bash
# Counter passes over the conv4 filter-gradient kernel alone (counters only next to --kernel-trace).  usage: wgrad_pmc.sh <tag>
R=$(pwd); OUT=$R/gpurun_out/wgrad_pmc_$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o pmc -- python $R/tools/ubench/wgrad_run.py 6 > $OUT/g$i.log 2>&1 || echo "group $i failed"
done
cd $R
python tools/ubench/pmc_avg.py $OUT --match wgrad > $OUT/summary.txt 2>&1
cat $OUT/summary.txt

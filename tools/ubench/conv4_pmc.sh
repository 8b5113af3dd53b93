#!/bin/bash
# Counter passes over the conv4 layer alone (run on an MI355X box from the repo root): L2 hit rate, L1 <-> L2 request
# latency, LDS conflicts / FIFO stalls, texture-path busy / stall cycles, wave-state cycles.  Counters only next to
# --kernel-trace.  usage: conv4_pmc.sh <tag>   (environment selects the kernel variant)
R=$(pwd); OUT=$R/gpurun_out/conv4_pmc_$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "TA_TA_BUSY_sum TD_TD_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o pmc -- python $R/tools/ubench/conv4_run.py 6 > $OUT/g$i.log 2>&1 || echo "group $i failed"
done
cd $R
python tools/ubench/pmc_avg.py $OUT --match os_ > $OUT/summary.txt 2>&1
cat $OUT/summary.txt

#!/bin/bash
OUT=/root/repo/gpurun_out/modepmc; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for set in "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUBBLE_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "GRBM_EA_BUSY GRBM_TC_BUSY" \
           "TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_STREAMING_REQ_sum TCC_WRITEBACK_sum" \
           "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o m -- python3 /root/repo/tools/mode_pmc.py > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  echo "== set $i: $set"; python3 /root/repo/tools/mode_pmc_report.py $f 2>&1 | head -12
done

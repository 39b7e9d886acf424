#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out/pmc; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/pmc/counters.txt 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "FETCH_SIZE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr" \
           "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  YOLO2_KSPLIT_BLOCKS=${KS:-0} timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc/p$i -o r -- python $R/scripts/one_layer.py conv2 conv5 conv8 conv13 conv18 > $R/gpurun_out/pmc/p$i.log 2>&1
  tail -2 $R/gpurun_out/pmc/p$i.log | cut -c1-200
done
find $R/gpurun_out/pmc -name "*.csv" | head -20

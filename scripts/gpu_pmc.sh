#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out/pmc; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc/p*
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc/p$i -o r -- python $R/scripts/one_layer.py conv2 conv5 conv8 conv13 conv18 conv20 > $R/gpurun_out/pmc/p$i.log 2>&1
  tail -2 $R/gpurun_out/pmc/p$i.log | cut -c1-200
done
python $R/scripts/pmc_summary.py $R/gpurun_out/pmc > $R/gpurun_out/pmc_summary.md 2>&1
cat $R/gpurun_out/pmc_summary.md
find $R/gpurun_out/pmc -name "*kernel_trace.csv" -delete

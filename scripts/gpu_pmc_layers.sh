#!/bin/bash
# SQ counters of the forward convolution + filter gradient of the layers named on the command line (default conv1), three counter sets in separate passes;
# table by scripts/pmc_summary.py with the MFMA-busy normalisation given in $NORM (default 120.5: the 16-waves-per-CU ratio of scripts/gpu_pmc.sh's calibration)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/pmc_layers; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
LAYERS="${*:-conv1}"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o r -- python $R/scripts/one_layer.py $LAYERS > $OUT/p$i.log 2>&1
  tail -1 $OUT/p$i.log | cut -c1-160
done
python $R/scripts/pmc_summary.py $OUT ${NORM:-120.5} > $R/gpurun_out/pmc_layers_summary.md 2>&1
cat $R/gpurun_out/pmc_layers_summary.md
find $OUT -name "*kernel_trace.csv" -delete

#!/bin/bash
# SQ counters of the conv kernels (forward igemm + wgrad of 6 layers), with the MFMA-busy normalisation calibrated on a pure MFMA loop
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out/pmc; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc/p* $R/gpurun_out/pmc/cal
mkdir -p $R/scripts/experiments/build
[ -x $R/scripts/experiments/build/mfma_peak ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/scripts/experiments/mfma_peak.hip -o $R/scripts/experiments/build/mfma_peak 2>/dev/null
timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc/cal -o r -- $R/scripts/experiments/build/mfma_peak > $R/gpurun_out/pmc/cal.log 2>&1
python - <<PY | tee $R/gpurun_out/pmc/calibration.txt
import csv, glob
rows = {}
for f in glob.glob('$R/gpurun_out/pmc/cal/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        e = rows.setdefault(int(r['Dispatch_Id']), {'grid': int(r['Grid_Size']), 'wg': int(r['Workgroup_Size'])})
        e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
print('pure MFMA loop (scripts/experiments/mfma_peak.hip): SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE per dispatch')
for k in sorted(rows):
    e = rows[k]
    print('  waves/CU %2d: MFMA_BUSY %.3e  GUI_ACTIVE %.3e  ratio %.3f   SQ_BUSY_CYCLES %.3e  INSTS_MFMA %.3e (x32 = %.3e)' % (
        e['grid'] // 64 // 256, e.get('SQ_VALU_MFMA_BUSY_CYCLES', 0), e.get('GRBM_GUI_ACTIVE', 1), e.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / e.get('GRBM_GUI_ACTIVE', 1),
        e.get('SQ_BUSY_CYCLES', 0), e.get('SQ_INSTS_MFMA', 0), 32 * e.get('SQ_INSTS_MFMA', 0)))
PY
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc/p$i -o r -- python $R/scripts/one_layer.py conv2 conv5 conv8 conv13 conv18 conv20 > $R/gpurun_out/pmc/p$i.log 2>&1
  tail -2 $R/gpurun_out/pmc/p$i.log | cut -c1-200
done
NORM=$(python -c "
import re
v=[float(m) for m in re.findall(r'waves/CU 16:.*?ratio ([0-9.]+)', open('$R/gpurun_out/pmc/calibration.txt').read())]
print('%.3f' % (sum(v)/len(v)) if v else '4.0')")
echo "MFMA_NORM (16 waves/CU pure-MFMA ratio) = $NORM" | tee -a $R/gpurun_out/pmc/calibration.txt
python $R/scripts/pmc_summary.py $R/gpurun_out/pmc $NORM > $R/gpurun_out/pmc_summary.md 2>&1
cat $R/gpurun_out/pmc_summary.md
find $R/gpurun_out/pmc -name "*kernel_trace.csv" -delete

#!/bin/bash
# timing ablations of the row-of-taps filter-gradient kernel (experiments library: bash scripts/experiments_build.sh w3; results wrong by design)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
EXP=$R/yolo_tf_amd/csrc/libyolo2hip_exp.so
{ hostname; /opt/rocm/bin/rocm-smi --showuniqueid 2>/dev/null | grep -i "unique"; } > gpurun_out/w3_box.txt 2>&1
: > gpurun_out/w3_abl.log
for a in ${ABLS:-0 1 2 3 4 8 16 7}; do
  echo "== YOLO2_W3_ABL=$a (1 no MFMA, 2 no fragment reads, 4 no DMA in the loop, 8 DMA without address arithmetic, 16 no barrier)" >> gpurun_out/w3_abl.log
  YOLO2_LIB_PATH=$EXP YOLO2_W3_ABL=$a VARIANTS=${VARIANTS:-0} LAYERS=${LAYERS:-conv5,conv8,conv13,conv20} timeout 120 python scripts/wgrad_ab.py abl$a 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" >> gpurun_out/w3_abl.log
done
cat gpurun_out/w3_abl.log

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
: > gpurun_out/tune.log
YOLO2_IGEMM_T256=16 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "conv_forward or conv_dgrad or conv_bn" 2>&1 | tail -3 | tee -a gpurun_out/tune.log
for cfg in "YOLO2_IGEMM_T256=0" "YOLO2_IGEMM_T256=128" "YOLO2_IGEMM_T256=64" "YOLO2_IGEMM_T256=32"; do
  env $cfg python scripts/conv_bench.py "$cfg" 2>/dev/null | grep -v amdgpu.ids >> gpurun_out/tune.log
done
cat gpurun_out/tune.log

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
: > gpurun_out/tune.log
YOLO2_WGRAD_64X128=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "wgrad" < /dev/null 2>&1 | tail -3 | tee -a gpurun_out/tune.log
for cfg in "YOLO2_WGRAD_64X128=0" "YOLO2_WGRAD_64X128=1" "YOLO2_WGRAD_64X128=1 YOLO2_WGRAD_BLOCKS=512"; do
  env $cfg timeout 200 python scripts/conv_bench.py "$cfg" 2>/dev/null < /dev/null | grep -v amdgpu.ids >> gpurun_out/tune.log
done
cat gpurun_out/tune.log

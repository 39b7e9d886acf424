#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
: > gpurun_out/tune.log
for cfg in "YOLO2_WGRAD_SMALL_BELOW=0" "YOLO2_WGRAD_SMALL_BELOW=9" "YOLO2_WGRAD_SMALL_BELOW=33" "YOLO2_WGRAD_SMALL_BELOW=80"; do
  env $cfg python scripts/conv_bench.py "$cfg" 2>/dev/null | grep -v amdgpu.ids >> gpurun_out/tune.log
done
cat gpurun_out/tune.log

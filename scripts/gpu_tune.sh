#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
: > gpurun_out/tune.log
for cfg in "YOLO2_WGRAD_BLOCKS=0" "YOLO2_WGRAD_BLOCKS=288" "YOLO2_WGRAD_BLOCKS=384" "YOLO2_WGRAD_BLOCKS=768"; do
  env $cfg timeout 200 python scripts/conv_bench.py "$cfg" 2>/dev/null < /dev/null | grep -v amdgpu.ids >> gpurun_out/tune.log
done
cat gpurun_out/tune.log

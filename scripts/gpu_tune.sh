#!/bin/bash
# Template for one-off kernel experiments on the GPU box (always under `timeout`, stdin closed):
#   per-layer table under a list of env settings -> gpurun_out/tune.log
# usage: gpurun -- 'bash scripts/gpu_tune.sh "YOLO2_X=0" "YOLO2_X=1" < /dev/null'
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
: > gpurun_out/tune.log
for cfg in "$@"; do
  env $cfg timeout 200 python scripts/conv_bench.py "$cfg" 2>/dev/null < /dev/null | grep -v amdgpu.ids >> gpurun_out/tune.log
done
cat gpurun_out/tune.log

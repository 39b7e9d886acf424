#!/bin/bash
# Kernel iteration call: the conv parity tests, then the per-layer microbench under each configuration line of $1 (default below).
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "${TESTK:-conv or wgrad}" -p no:cacheprovider 2>&1 | tail -5
CFG=${1:-scripts/kernel_cfgs.txt}
: > gpurun_out/conv_bench.log
while IFS= read -r cfg; do
  [ -z "$cfg" ] && continue
  env $cfg timeout 300 python scripts/conv_bench.py "$cfg" 2>/dev/null | tee -a gpurun_out/conv_bench.log
done < $CFG

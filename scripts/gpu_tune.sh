#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
make -C oracle >/dev/null 2>&1
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "wgrad" 2>&1 | tail -2
{
for cfg in "1024 0" "1024 1" "512 1" "256 1" "2048 1"; do set -- $cfg
YOLO2_WGRAD_BLOCKS=$1 YOLO2_WGRAD_REMAP=$2 python scripts/conv_bench.py "wgrad blocks$1 remap$2"
done
} > gpurun_out/conv_bench.log 2>&1
grep "totals" gpurun_out/conv_bench.log

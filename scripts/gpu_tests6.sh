#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
make -C oracle >/dev/null 2>&1
timeout 160 python -m pytest tests/test_kernels_gpu.py -k "tap_fused and (3x5x7 or 16x26x26x256x136)" -q -p no:cacheprovider --durations=0 -o faulthandler_timeout=3 2>&1 | grep -v "^  File \"/usr" | tail -150 > gpurun_out/t6_tap.log; tail -60 gpurun_out/t6_tap.log

#!/bin/bash
mkdir -p gpurun_out
YOLO2_IGEMM_TAP_MIN_STEPS=0 timeout 900 python -m pytest tests/test_kernels_gpu.py -k "tap_fused" -x -q -m gpu 2>&1 | grep -v "^$" | tail -30 | cut -c1-330 | tee gpurun_out/new_tests.log
LAYERS=conv5,conv8,conv13,conv18,conv20 YOLO2_IGEMM_TAP=0 timeout 300 python scripts/conv_bench.py tap0 2>&1 | tail -8 | tee gpurun_out/conv_tap0.txt
LAYERS=conv5,conv8,conv13,conv18,conv20 YOLO2_IGEMM_TAP_MIN_STEPS=0 timeout 300 python scripts/conv_bench.py tap1 2>&1 | tail -8 | tee gpurun_out/conv_tap1.txt

#!/bin/bash
mkdir -p gpurun_out
YOLO2_IGEMM_TAP_MIN_STEPS=0 YOLO2_IGEMM_TAP_MIN_SHARE=12 timeout 900 python -m pytest tests/test_kernels_gpu.py -k "tap_fused" -x -q -m gpu 2>&1 | grep -v "^$" | tail -5 | cut -c1-330 | tee gpurun_out/new_tests.log
LAYERS=conv13,conv18,conv20 YOLO2_IGEMM_TAP_MIN_STEPS=0 YOLO2_IGEMM_TAP_MIN_SHARE=12 timeout 300 python scripts/conv_bench.py tap1 2>&1 | tail -6 | tee gpurun_out/conv_tap1.txt
bash scripts/gpu_pmc.sh > gpurun_out/pmc_run.log 2>&1; tail -16 gpurun_out/pmc_summary.md | cut -c1-250

#!/bin/bash
# Scratch call used while iterating on one change: edit the two lines below, then `gpurun -- 'bash scripts/gpu_new.sh'`.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "${TESTK:-conv}" -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/new_tests.log
LAYERS=${LAYERS:-conv5,conv8,conv13,conv18,conv20} timeout 300 python scripts/conv_bench.py scratch 2>&1 | tail -8 | tee gpurun_out/conv_scratch.txt

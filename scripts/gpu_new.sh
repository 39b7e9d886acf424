#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -k "conv_forward or bn_fused or conv0 or first" -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/new_tests.log
LAYERS=conv0,conv1 timeout 300 python scripts/conv_bench.py first_wide 2>&1 | grep "^conv" | tee gpurun_out/conv_first.txt

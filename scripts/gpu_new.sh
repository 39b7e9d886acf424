#!/bin/bash
mkdir -p gpurun_out
YOLO2_IGEMM_NW4=1 timeout 900 python -m pytest tests/test_bench_shapes_gpu.py -k "fwd or dgrad" -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/new_tests.log
LAYERS=conv2,conv5,conv8 YOLO2_IGEMM_NW4=0 timeout 300 python scripts/conv_bench.py nw8 2>&1 | tail -6 | tee gpurun_out/conv_nw8.txt
LAYERS=conv2,conv5,conv8 YOLO2_IGEMM_NW4=1 timeout 300 python scripts/conv_bench.py nw4 2>&1 | tail -6 | tee gpurun_out/conv_nw4.txt

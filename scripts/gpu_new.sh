#!/bin/bash
mkdir -p gpurun_out
YOLO2_WGRAD_SMALL_BKP64=1 timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -k "wgrad" -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/new_tests.log
for v in 0 1; do
  YOLO2_WGRAD_SMALL_BKP64=$v LAYERS=conv2,conv3,conv5,conv6,conv9,conv14 timeout 300 python scripts/conv_bench.py small64_$v 2>&1 | grep "^conv\|totals" | cut -c1-70,95- | tee -a gpurun_out/conv_scratch.txt
done

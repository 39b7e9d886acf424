#!/bin/bash
# Scratch call used while iterating on one change: edit the lines below, then `gpurun -- 'bash scripts/gpu_new.sh'`.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -k "${TESTK:-fwd or dgrad or conv_forward or conv_dgrad or bn_fused}" -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/new_tests.log
for L in scripts/experiments/build/lib_base.so yolo_tf_amd/csrc/libyolo2hip.so; do
  YOLO2_LIB_PATH=$PWD/$L LAYERS=conv2,conv5,conv8,conv13 timeout 300 python scripts/conv_bench.py $(basename $L) 2>&1 | tail -7 | tee -a gpurun_out/conv_scratch.txt
done

#!/bin/bash
# the tests touched by the change under development (fast iteration), then an A/B of the bench line
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -k "dgrad_bn or bn_fused or conv_dgrad" -x -q -s -m gpu 2>&1 | grep -v "^$" | tail -40 | cut -c1-330 | tee gpurun_out/new_tests.log
timeout 900 python -m pytest tests/test_network_gpu.py tests/test_yolo1_gpu.py -x -q -m gpu -k "layerwise or train_step or matches_oracle or multi_scale" 2>&1 | tail -15 | tee -a gpurun_out/new_tests.log
bash scripts/bench_ab.sh YOLO2_STATS_TAIL=0 YOLO2_STATS_TAIL=1 YOLO2_STATS_TAIL=0 YOLO2_STATS_TAIL=1

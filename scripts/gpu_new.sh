#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_network_gpu.py -k "overlapped or train_step or multi_scale or tensorflow" -x -q -s -m gpu 2>&1 | grep -v "^$" | tail -12 | cut -c1-300 | tee gpurun_out/new_tests.log
bash scripts/bench_ab.sh YOLO2_OVERLAP_OPTIMIZER=0 YOLO2_OVERLAP_OPTIMIZER=1 YOLO2_OVERLAP_OPTIMIZER=0 YOLO2_OVERLAP_OPTIMIZER=1

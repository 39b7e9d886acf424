#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -k "wgrad" -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/new_tests.log
timeout 300 python scripts/conv_bench.py auto64 2>&1 | grep "^conv\|totals" | cut -c1-70,95- | tee -a gpurun_out/conv_scratch.txt

#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -k "wgrad" -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/new_tests.log
timeout 300 python scripts/conv_bench.py wgrad_imm 2>&1 | tail -16 | tee gpurun_out/conv_wgrad_imm.txt

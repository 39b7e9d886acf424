#!/bin/bash
# the tests touched by the change under development (fast iteration), then the conv microbenchmark
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "dgrad_bn or bn_fused or conv_dgrad" -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/new_tests.log
timeout 900 python -m pytest tests/test_network_gpu.py -x -q -m gpu -k "layerwise or train_step or matches_oracle" 2>&1 | tail -15 | tee -a gpurun_out/new_tests.log
timeout 600 python bench.py --no-cpu-baseline --no-detect 2>&1 | tail -1 | cut -c1-1500 | tee gpurun_out/bench_quick.json
YOLO2_FUSE_BN_BWD=0 timeout 600 python bench.py --no-cpu-baseline --no-detect 2>&1 | tail -1 | cut -c1-300 | tee gpurun_out/bench_quick_nofuse.json

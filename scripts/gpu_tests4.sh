#!/bin/bash
# round 4: the GPU tests the earlier (time-capped) full run did not reach, with durations
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
make -C oracle >/dev/null 2>&1
timeout 1000 python -m pytest tests/test_kernels_gpu.py -k "tap_fused or nms or shift_is_read_only or dgrad_bn" -q -p no:cacheprovider --timeout 300 --durations=8 2>&1 | tail -25 > gpurun_out/t4_kernels.log; tail -14 gpurun_out/t4_kernels.log
timeout 1200 python -m pytest tests/test_network_gpu.py tests/test_reference_pins_gpu.py tests/test_streamk_occupied_gpu.py tests/test_rccl_gpu.py tests/test_comm_gpu.py -q -p no:cacheprovider --timeout 600 --durations=12 -x 2>&1 | tail -40 > gpurun_out/t4_network.log; tail -30 gpurun_out/t4_network.log

#!/bin/bash
# round 4: the tests added after the last targeted run + where a small tap-fused case spends its seconds on the GPU box's host
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
make -C oracle >/dev/null 2>&1
nproc > gpurun_out/t5_host.txt; python -c "import torch; print('torch threads', torch.get_num_threads())" >> gpurun_out/t5_host.txt 2>&1
timeout 170 python -m cProfile -o /tmp/tap.prof -m pytest tests/test_kernels_gpu.py -k "tap_fused and 3x5x7" -q -p no:cacheprovider --durations=10 2>&1 | tail -16 > gpurun_out/t5_tap_small.log
python -c "
import pstats; pstats.Stats('/tmp/tap.prof').sort_stats('cumtime').print_stats(45)" 2>&1 | cut -c1-180 | tail -60 >> gpurun_out/t5_tap_small.log; cat gpurun_out/t5_tap_small.log
timeout 400 python -m pytest tests/test_rccl_gpu.py "tests/test_network_gpu.py::test_multi_scale_batch8_teacher_forced_forward" -q -p no:cacheprovider --timeout 350 --durations=5 -s 2>&1 | tail -25 > gpurun_out/t5_new.log; cat gpurun_out/t5_new.log

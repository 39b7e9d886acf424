#!/bin/bash
# round 4, run 1: ping-pong kernel -- parity of every variant, then the per-layer A/B table at batch 16 and 8
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
make -C oracle >/dev/null 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "tap_fused" -q -p no:cacheprovider --timeout 180 2>&1 | tail -40 > gpurun_out/pp1_pytest.log
tail -15 gpurun_out/pp1_pytest.log
B=16 timeout 600 python scripts/pp_sweep.py > gpurun_out/pp1_sweep_b16.log 2>&1; cat gpurun_out/pp1_sweep_b16.log
B=8 timeout 600 python scripts/pp_sweep.py > gpurun_out/pp1_sweep_b8.log 2>&1; cat gpurun_out/pp1_sweep_b8.log

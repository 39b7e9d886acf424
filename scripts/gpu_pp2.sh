#!/bin/bash
# round 4, run 2: ping-pong kernel SCHED variants + timing ablations, per layer; two quick regression tests
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
make -C oracle >/dev/null 2>&1
C="per-tap:0:0:0,tap-r2:1:0:0,s0:2:1:0,s2:2:1:2,s3:2:1:3,s4:2:1:4,s10:2:1:10,s12:2:1:12,s18:2:1:18,s26:2:1:26,bwdtpl:2:1:34"
B=16 CONFIGS=$C timeout 600 python scripts/pp_sweep.py > gpurun_out/pp2_sched_b16.log 2>&1; cat gpurun_out/pp2_sched_b16.log
A="s2:2:1:2,noMFMA:2:1:66,noREAD:2:1:130,noDMA:2:1:258,noMFMAnoREAD:2:1:194,noREADnoDMA:2:1:386,noMFMAnoDMA:2:1:322,skeleton:2:1:450"
B=16 LAYERS=conv8,conv18,conv20 WHAT=fwd+stats CONFIGS=$A timeout 600 python scripts/pp_sweep.py > gpurun_out/pp2_abl_b16.log 2>&1; cat gpurun_out/pp2_abl_b16.log
T="per-tap:0:0:0,s2sk:2:1:2,s2tile:2:2:2,s10tile:2:2:10"
B=16 CONFIGS=$T timeout 600 python scripts/pp_sweep.py > gpurun_out/pp2_tile_b16.log 2>&1; cat gpurun_out/pp2_tile_b16.log
B=8 CONFIGS=$T timeout 600 python scripts/pp_sweep.py > gpurun_out/pp2_tile_b8.log 2>&1; cat gpurun_out/pp2_tile_b8.log
timeout 300 python -m pytest tests/test_kernels_gpu.py -k "shift_is_read_only or nms" -q -p no:cacheprovider --timeout 120 2>&1 | tail -5 > gpurun_out/pp2_pytest.log; cat gpurun_out/pp2_pytest.log

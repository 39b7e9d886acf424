#!/bin/bash
# round 4, run 3: ORDER 5 (pixel fragments of step s+1 read behind the MFMAs of step s), grids: stream-K 256, one per tile, tiles x 2 / x 3
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
make -C oracle >/dev/null 2>&1
C="per-tap:0:0:0,s2:2:1:2,s5:2:1:5,s13:2:1:13,s5tile:2:2:5,s5x2:2:-2:5,s5x3:2:-3:5,s5x4:2:-4:5"
B=16 CONFIGS=$C timeout 600 python scripts/pp_sweep.py > gpurun_out/pp3_b16.log 2>&1; cat gpurun_out/pp3_b16.log
B=8 CONFIGS=$C timeout 600 python scripts/pp_sweep.py > gpurun_out/pp3_b8.log 2>&1; cat gpurun_out/pp3_b8.log
A="s5:2:1:5,noMFMA:2:1:69,noREAD:2:1:133,noDMA:2:1:261,noMFMAnoREAD:2:1:197,noREADnoDMA:2:1:389,noMFMAnoDMA:2:1:325,skeleton:2:1:453"
B=16 LAYERS=conv8,conv20 WHAT=fwd+stats CONFIGS=$A timeout 600 python scripts/pp_sweep.py > gpurun_out/pp3_abl_b16.log 2>&1; cat gpurun_out/pp3_abl_b16.log
timeout 300 python -m pytest tests/test_kernels_gpu.py -k "shift_is_read_only" -q -p no:cacheprovider --timeout 120 2>&1 | tail -25 > gpurun_out/pp3_pytest.log; cat gpurun_out/pp3_pytest.log

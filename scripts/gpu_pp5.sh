#!/bin/bash
# round 4, run 5: ORDER 6 (single-phase software pipeline), fixed-cost breakdown of the ping-pong kernel, atomic-add rate, measured pin errors
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
make -C oracle >/dev/null 2>&1
# (at the time this ran, pp_experiments_build.sh rebuilt libyolo2hip.so in place; it now builds libyolo2hip_exp.so, selected with YOLO2_LIB_PATH)
bash scripts/pp_experiments_build.sh || exit 1; export YOLO2_LIB_PATH=$PWD/yolo_tf_amd/csrc/libyolo2hip_exp.so
C="per-tap:0:0:0,s2:2:1:2,s6:2:1:6,s14:2:1:14,s2tile:2:2:2,s6tile:2:2:6"
B=16 CONFIGS=$C timeout 600 python scripts/pp_sweep.py > gpurun_out/pp5_b16.log 2>&1; cat gpurun_out/pp5_b16.log
B=8 CONFIGS=$C timeout 600 python scripts/pp_sweep.py > gpurun_out/pp5_b8.log 2>&1; cat gpurun_out/pp5_b8.log
A="s6:2:1:6,noMFMA:2:1:70,noREAD:2:1:134,noDMA:2:1:262,noMFMAnoREAD:2:1:198,noREADnoDMA:2:1:390,noMFMAnoDMA:2:1:326,skeleton:2:1:454"
B=16 LAYERS=conv8,conv20 WHAT=fwd+stats CONFIGS=$A timeout 600 python scripts/pp_sweep.py > gpurun_out/pp5_abl6_b16.log 2>&1; cat gpurun_out/pp5_abl6_b16.log
S="s2:2:1:2,skel:2:1:450,skel-noepi:2:1:962,skel-nohand:2:1:1474,skel-neither:2:1:1986,full-noepi:2:1:514,full-nohand:2:1:1026,tile:2:2:2,tile-skel:2:2:450,tile-skel-noepi:2:2:962"
B=16 LAYERS=conv8,conv13,conv18 WHAT=fwd+stats,dgrad CONFIGS=$S timeout 600 python scripts/pp_sweep.py > gpurun_out/pp5_fixed_b16.log 2>&1; cat gpurun_out/pp5_fixed_b16.log
scripts/experiments/build/atomic_rate > gpurun_out/pp5_atomic_rate.log 2>&1; cat gpurun_out/pp5_atomic_rate.log
timeout 600 python -m pytest tests/test_reference_pins_gpu.py -q -s -p no:cacheprovider --timeout 300 2>&1 | grep -E "MEASURED|passed|failed|Error" > gpurun_out/pp5_pins.log; cat gpurun_out/pp5_pins.log
python yolo_tf_amd/csrc/build.py --force > /dev/null 2>&1

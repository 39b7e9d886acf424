#!/bin/bash
# round 4, run 7: order 7 = filter fragments of step s+1 read behind the last MFMA of step s
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
hostname > gpurun_out/pp7_box.txt; /opt/rocm/bin/rocm-smi --showserial --showuniqueid 2>/dev/null | grep -i "GPU\[0\]" >> gpurun_out/pp7_box.txt
C="per-tap:0:0:0,pp2:2:0:2,pp7:2:0:7,pp2-tile:2:2:2,pp7-tile:2:2:7"
B=16 CONFIGS=$C timeout 600 python scripts/pp_sweep.py > gpurun_out/pp7_b16.log 2>&1; cat gpurun_out/pp7_b16.log
B=8 LAYERS=conv8,conv13,conv20 CONFIGS="per-tap:0:0:0,pp2:2:0:2,pp7:2:0:7" timeout 600 python scripts/pp_sweep.py > gpurun_out/pp7_b8.log 2>&1; cat gpurun_out/pp7_b8.log
PP_SCHED=7 python scripts/conv_bench.py "r04c" > gpurun_out/pp7_conv_bench.log 2>&1; tail -6 gpurun_out/pp7_conv_bench.log

#!/bin/bash
# round 4, run 6: prologue DMA before the index arithmetic, deferred publication of parked tiles, specialised staging loop
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
make -C oracle >/dev/null 2>&1
C="per-tap:0:0:0,pp-rule:2:0:2,pp-sk:2:1:2,pp-tile:2:2:2"
B=16 CONFIGS=$C timeout 600 python scripts/pp_sweep.py > gpurun_out/pp6_b16.log 2>&1; cat gpurun_out/pp6_b16.log
B=8 CONFIGS=$C timeout 600 python scripts/pp_sweep.py > gpurun_out/pp6_b8.log 2>&1; cat gpurun_out/pp6_b8.log
python scripts/conv_bench.py "r04b" > gpurun_out/pp6_conv_bench.log 2>&1; tail -18 gpurun_out/pp6_conv_bench.log

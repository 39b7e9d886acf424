#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o run -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/prof.log 2>&1
grep '"metric"' $R/gpurun_out/prof.log | cut -c1-200
python $R/scripts/prof_summary.py $R/gpurun_out/prof 5 > $R/gpurun_out/prof_summary.md
head -40 $R/gpurun_out/prof_summary.md

#!/bin/bash
# kernel trace of the product configuration + single-stream, with a per-launch listing of the last step
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-detect"
rm -rf $R/gpurun_out/prof $R/gpurun_out/prof1s
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o run -- $B > $R/gpurun_out/prof.log 2>&1
grep '"metric"' $R/gpurun_out/prof.log | cut -c1-200
python $R/scripts/prof_summary.py $R/gpurun_out/prof 5 > $R/gpurun_out/prof_summary.md
YOLO2_OVERLAP_WGRAD=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof1s -o run -- $B > $R/gpurun_out/prof1s.log 2>&1
grep '"metric"' $R/gpurun_out/prof1s.log | cut -c1-200
python $R/scripts/prof_summary.py $R/gpurun_out/prof1s 5 > $R/gpurun_out/prof1s_summary.md
python $R/scripts/prof_step_listing.py $R/gpurun_out/prof1s > $R/gpurun_out/prof1s_last_step.txt
find $R/gpurun_out/prof $R/gpurun_out/prof1s -name "*kernel_trace.csv" -size +20M -delete
find $R/gpurun_out/prof $R/gpurun_out/prof1s -name "*.db" -size +20M -delete

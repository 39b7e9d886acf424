#!/bin/bash
# kernel trace (product mode + single stream) at a chosen batch / class count:  gpu_prof_b.sh <batch> <names> <tag>
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
BT=${1:-8}; NM=${2:-80}; TAG=${3:-b${BT}}
B="python $R/bench.py --steps 6 --warmup 2 --batch $BT --names $NM --no-cpu-baseline --no-kernel-timer --no-detect"
for mode in ov 1s; do
  D=$R/gpurun_out/prof_${TAG}_$mode; rm -rf $D
  if [ $mode = 1s ]; then export YOLO2_OVERLAP_WGRAD=0; else unset YOLO2_OVERLAP_WGRAD; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d $D -o run -- $B > $D.log 2>&1
  grep '"metric"' $D.log | cut -c1-200
  python $R/scripts/prof_summary.py $D 5 > $R/gpurun_out/prof_${TAG}_${mode}_summary.md
  [ $mode = 1s ] && python $R/scripts/prof_step_listing.py $D > $R/gpurun_out/prof_${TAG}_1s_last_step.txt
  find $D -name "*kernel_trace.csv" -size +20M -delete; find $D -name "*.db" -size +20M -delete
done

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/profd
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/profd -o run -- python $R/scripts/detect_prof.py 256 > $R/gpurun_out/profd.log 2>&1
grep "detect batch" $R/gpurun_out/profd.log
f=$(find $R/gpurun_out/profd -name "*kernel_stats.csv" | head -1)
head -25 $f | cut -c1-200 > $R/gpurun_out/profd_stats.txt; cat $R/gpurun_out/profd_stats.txt
find $R/gpurun_out/profd -name "*kernel_trace.csv" -delete; find $R/gpurun_out/profd -name "*.db" -delete

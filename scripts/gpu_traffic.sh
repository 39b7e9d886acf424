#!/bin/bash
# Final-state profiles: kernel trace (product config + single-stream), then HBM traffic counters in separate PMC passes.
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-detect --no-f32"
rm -rf $R/gpurun_out/prof $R/gpurun_out/prof1s $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o run -- $B > $R/gpurun_out/prof.log 2>&1
grep '"metric"' $R/gpurun_out/prof.log | cut -c1-200
python $R/scripts/prof_summary.py $R/gpurun_out/prof 6 > $R/gpurun_out/prof_summary.md
python $R/scripts/prof_roofline.py $R/gpurun_out/prof 6 > $R/gpurun_out/prof_roofline.txt
python -c "
import json,sys
for l in open('$R/gpurun_out/prof.log'):
    if l.startswith('{\"metric\"'):
        r=json.loads(l)['roofline']; print('bench.py under rocprofv3: roofline.avg_launch_ms %.5f over %d launches, frac %.4f' % (r['avg_launch_ms'], r['launches'], r['frac']))
" >> $R/gpurun_out/prof_roofline.txt
cat $R/gpurun_out/prof_roofline.txt
YOLO2_OVERLAP_WGRAD=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof1s -o run -- $B > $R/gpurun_out/prof1s.log 2>&1
grep '"metric"' $R/gpurun_out/prof1s.log | cut -c1-200
python $R/scripts/prof_summary.py $R/gpurun_out/prof1s 6 > $R/gpurun_out/prof1s_summary.md
python $R/scripts/prof_step_listing.py $R/gpurun_out/prof1s > $R/gpurun_out/prof1s_last_step.txt
[ "$1" = notraffic ] && exit 0      # (scripts/gpu_evidence.sh trace: kernel traces only)
B2="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-detect --no-f32"
YOLO2_OVERLAP_WGRAD=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o r -- $B2 > $R/gpurun_out/pmc_fetch.log 2>&1
YOLO2_OVERLAP_WGRAD=0 timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o r -- $B2 > $R/gpurun_out/pmc_write.log 2>&1
python $R/scripts/pmc_traffic.py $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write > $R/gpurun_out/traffic_summary.md 2>&1
python $R/scripts/prof_roofline.py $R/gpurun_out/prof 6 $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write | tail -1 | tee -a $R/gpurun_out/prof_roofline.txt
tail -3 $R/gpurun_out/traffic_summary.md
# keep the merged-back payload small
find $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/prof $R/gpurun_out/prof1s -name "*kernel_trace.csv" -size +20M -delete
find $R/gpurun_out/prof $R/gpurun_out/prof1s -name "*.db" -size +30M -delete

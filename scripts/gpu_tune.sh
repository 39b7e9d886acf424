#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
: > $R/gpurun_out/tune.log
for v in 1024; do
  rm -rf $R/gpurun_out/proft
  YOLO2_POOL_REDUCE_BLOCKS=$v YOLO2_OVERLAP_WGRAD=0 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/proft -o run -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-detect < /dev/null > $R/gpurun_out/proft.log 2>&1
  echo "== YOLO2_POOL_REDUCE_BLOCKS=$v" >> $R/gpurun_out/tune.log
  timeout 60 python $R/scripts/prof_step_listing.py $R/gpurun_out/proft < /dev/null | grep -E "bn_pool_bwd_reduce|last training step" | awk "{print \$2, \$3, \$4}" | head -8 >> $R/gpurun_out/tune.log
done
rm -rf $R/gpurun_out/proft
cat $R/gpurun_out/tune.log

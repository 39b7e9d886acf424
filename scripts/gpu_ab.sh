#!/bin/bash
# A/B of engine/kernels knobs on one box: each line = env assignments; prints img/s
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
: > gpurun_out/ab.log
while IFS= read -r cfg; do
  [ -z "$cfg" ] && continue
  for rep in 1 2; do
    v=$(env $cfg python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-detect --no-kernel-timer 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))")
    echo "$cfg rep$rep: $v" | tee -a gpurun_out/ab.log
  done
done <<CFG
YOLO2_FUSE_IMAGE_BWD=1
YOLO2_FUSE_IMAGE_BWD=0
CFG

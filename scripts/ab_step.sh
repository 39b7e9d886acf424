#!/bin/bash
# Step-time A/B of library / environment configurations, alternating, in ONE gpurun call.  usage: bash scripts/ab_step.sh ROUNDS "NAME=ENV ENV ..." ...
#   e.g. bash scripts/ab_step.sh 4 "old=YOLO2_LIB_PATH=$PWD/profiles/baseline/libyolo2hip_r05.so YOLO2_LIB_BASELINE=1" "new=" "new_c64off=YOLO2_C64=0"
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
N=$1; shift
for i in $(seq 1 $N); do for cfg in "$@"; do
  name=${cfg%%=*}; envs=${cfg#*=}
  env $envs python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-detect --no-f32 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j.get('roofline') or {}
        print('%-12s run $i: %.3f ms/step %6.0f img/s   dominant launches %.2f us = %.3f of peak' % ('$name', j['ms_per_step'], j['value'], 1e3*(r.get('avg_launch_ms') or 0), r.get('frac') or 0))"
done; done

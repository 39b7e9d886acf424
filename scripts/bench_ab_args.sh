#!/bin/bash
# A/B of env switches with extra bench arguments: scripts/bench_ab_args.sh "<bench args>" "VAR=0" "VAR=1" ...
args=$1; shift
for cfg in "$@"; do
  env $cfg timeout 600 python bench.py $args --no-cpu-baseline --no-detect 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$args | $cfg', 'img/s %.0f  ms %.3f' % (d['value'], d['ms_per_step']))" | tee -a gpurun_out/bench_ab.txt
done

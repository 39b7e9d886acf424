#!/bin/bash
mkdir -p gpurun_out
for a in "--multiscale --no-cpu-baseline --no-detect" "--names 80 --batch 8 --no-cpu-baseline --no-detect" "--dtype f32 --steps 5 --warmup 2 --no-cpu-baseline --no-detect" "--batch 32 --steps 10 --no-cpu-baseline --no-detect"; do
  timeout 600 python bench.py $a 2>&1 | tail -1 | python -c "
import json,sys
l=sys.stdin.read()
try:
    d=json.loads(l); r=d.get('roofline') or {}
    print('$a ->', d['metric'], '%.0f %s  %.3f ms  frac %s  loss %s' % (d['value'], d['unit'], d['ms_per_step'], r.get('frac'), d.get('total_loss')))
except Exception as e:
    print('$a -> FAILED', l[-400:])
" | tee -a gpurun_out/bench_cfgs.txt
done

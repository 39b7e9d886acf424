#!/bin/bash
# A/B of env switches on one box: scripts/bench_ab.sh "VAR=0" "VAR=1" ...   (compact fields of the bench line)
mkdir -p gpurun_out
for cfg in "$@"; do
  env $cfg timeout 600 python bench.py --no-cpu-baseline --no-detect 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$cfg', 'img/s %.0f  ms %.3f  frac %.4f  avg_launch_us %.2f  fwd %.0f TF  dgrad %.0f TF' % (d['value'], d['ms_per_step'], r['frac'], r['avg_launch_ms']*1e3, r['forward_launches_tflops'], r['data_gradient_launches_tflops']))" | tee -a gpurun_out/bench_ab.txt
done

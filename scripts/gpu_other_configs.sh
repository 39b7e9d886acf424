#!/bin/bash
# the other BASELINE configurations + the batch fit on one box -> gpurun_out/other_configs.txt
mkdir -p gpurun_out; out=gpurun_out/other_configs.txt; rm -f $out
for a in "--batch 4" "--batch 8" "--batch 16" "--batch 32" "--batch 8 --names 80" "--batch 16 --dtype f32 --steps 10 --warmup 3" "--multiscale"; do
  python bench.py $a --no-cpu-baseline --no-detect --no-f32 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$a ->', '%.0f img/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" | tee -a $out
done
python - <<'PY' | tee -a gpurun_out/other_configs.txt
import re
t={}
for line in open('gpurun_out/other_configs.txt'):
    m=re.match(r'--batch (\d+) -> .* ([\d.]+) ms/step', line)
    if m: t[int(m.group(1))]=float(m.group(2))
xs=[b for b in (8,16,32) if b in t]
n=len(xs); sx=sum(xs); sy=sum(t[b] for b in xs); sxx=sum(b*b for b in xs); sxy=sum(b*t[b] for b in xs)
k=(n*sxy-sx*sy)/(n*sxx-sx*sx); c=(sy-k*sx)/n
print('# least squares over batch 8, 16, 32: t = %.3f ms + %.4f ms x images   (batch 4 measured %.3f, fit %.3f)' % (c, k, t.get(4, float('nan')), c+4*k))
print('# strong-scaling bound for 64 images over N GPUs (one GPU at batch 64 by the fit: %.2f ms): N=2 %.2f, N=4 %.2f, N=8 %.2f' % (c+64*k, (c+64*k)/(2*t[32]), (c+64*k)/(4*t[16]), (c+64*k)/(8*t[8])))
PY

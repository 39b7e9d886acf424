#!/bin/bash
R=$PWD; mkdir -p gpurun_out
timeout 300 python scripts/conv_bench.py final 2>&1 | grep -v amdgpu.ids > gpurun_out/conv_final.txt
cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/detprof
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/detprof -o run -- python $R/scripts/detect_prof.py 256 > $R/gpurun_out/detprof.log 2>&1
cd $R
python - <<PY > gpurun_out/detect_trace.md
import sys; sys.path.insert(0,'scripts')
from prof_summary import load, short
rows=load('gpurun_out/detprof')
nms=[i for i,r in enumerate(rows) if 'nms_kernel' in r[0]]
sel=rows[nms[-6]+1:nms[-1]+1]
agg={}
for n,s,e in sel:
    a=agg.setdefault(short(n)[:100],[0,0.0]); a[0]+=1; a[1]+=(e-s)/1e3
tot=sum(v[1] for v in agg.values())
print('# rocprofv3 --kernel-trace: batch-256 detect (scripts/detect_prof.py: standardise + forward with folded BN + decode + NMS on sparse scores), last 5 batches')
print('kernel time per batch %.2f ms, %d launches\n' % (tot/5/1e3, len(sel)//5))
print('| kernel | calls/batch | avg us | % |\n|---|---:|---:|---:|')
for n,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1]):
    print('| \`%s\` | %.1f | %.1f | %.1f |' % (n, c/5, t/c, 100*t/tot))
PY
find gpurun_out/detprof -name "*.db" -size +30M -delete
tail -3 gpurun_out/conv_final.txt; head -8 gpurun_out/detect_trace.md

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
{
for d in 0 9 10 11; do YOLO2_KSPLIT_BLOCKS=0 YOLO2_IGEMM_DBG=$d python scripts/conv_bench.py "dbg$d"; done
} > gpurun_out/conv_bench.log 2>&1
python - <<'PY'
txt=open('gpurun_out/conv_bench.log').read().split('\n')
runs={}; cur=None
for l in txt:
    if l.startswith('layer'): cur=l.split(')')[-1].strip(); runs[cur]={}
    elif l.startswith('conv') and cur: runs[cur][l.split()[0]]=l[8:].split()
names=list(runs)
print('fwd us:  %-8s'%'layer', ' '.join('%10s'%n for n in names))
for layer in runs[names[0]]:
    print('         %-8s'%layer, ' '.join('%10s'%runs[n][layer][0].split('|')[0] for n in names))
PY

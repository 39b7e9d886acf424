#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "wgrad" 2>&1 | tail -2
{
for cfg in "0 0" "1 0" "1 512" "1 1024" "1 2048"; do set -- $cfg
YOLO2_WGRAD_NW8=$1 YOLO2_WGRAD_BLOCKS=$2 python scripts/conv_bench.py "wgrad nw8=$1 blocks=$2"
done
} > gpurun_out/conv_bench.log 2>&1
python - <<'PY'
txt=open('gpurun_out/conv_bench.log').read().split('\n')
runs={}; cur=None
for l in txt:
    if l.startswith('layer'): cur=l.split(')')[-1].strip(); runs[cur]={}
    elif l.startswith('conv') and cur: runs[cur][l.split()[0]]=l[8:].split()
names=list(runs)
print('wgrad us: %-8s'%'layer', ' '.join('%22s'%n[6:] for n in names))
for layer in runs[names[0]]:
    print('          %-8s'%layer, ' '.join('%22s'%runs[n][layer][-1].split('|')[0] for n in names))
PY
grep totals gpurun_out/conv_bench.log

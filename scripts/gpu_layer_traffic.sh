#!/bin/bash
# time + fabric traffic of single conv layers (forward + filter gradient of scripts/one_layer.py), A/B over env settings:
#   gpu_layer_traffic.sh "conv18 conv20" "YOLO2_WGRAD_SINGLE_ORDER=0" "YOLO2_WGRAD_SINGLE_ORDER=1"
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
LAYERS=$1; shift
OUT=$R/gpurun_out/layer_traffic.txt; : > $OUT
for cfg in "$@"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    D=$R/gpurun_out/lt_$ctr; rm -rf $D
    env $cfg timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $D -o r -- python $R/scripts/one_layer.py $LAYERS > $D.log 2>&1
  done
  D=$R/gpurun_out/lt_trace; rm -rf $D
  env $cfg timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D -o r -- python $R/scripts/one_layer.py $LAYERS > $D.log 2>&1
  python - "$cfg" $R/gpurun_out/lt_FETCH_SIZE $R/gpurun_out/lt_WRITE_SIZE $D <<'PY' | tee -a $OUT
import csv, glob, os, sys
csv.field_size_limit(1 << 30)
cfg, fdir, wdir, tdir = sys.argv[1:5]
def pmc(d, name):
    rows = {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == name:
                e = rows.setdefault(int(r['Dispatch_Id']), [r['Kernel_Name'], 0.0])
                e[1] += float(r['Counter_Value'])
    return [rows[k] for k in sorted(rows)]
fe, wr = pmc(fdir, 'FETCH_SIZE'), pmc(wdir, 'WRITE_SIZE')
tr = []
for f in glob.glob(os.path.join(tdir, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        tr.append((int(r['Start_Timestamp']), r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
tr.sort()
tr = [t for t in tr if 'conv' in t[1]]
fe = [t for t in fe if 'conv' in t[0]]; wr = [t for t in wr if 'conv' in t[0]]
print('## %s' % cfg)
print('| kernel | us | fetch MB (x2) | write MB |')
n = min(len(fe), len(wr), len(tr))
seen = {}
for i in range(n):
    key = (fe[i][0][:70], i % 2)
    seen.setdefault(key, []).append((tr[i][2], fe[i][1] * 2048 / 1e6, wr[i][1] * 1024 / 1e6))
for (k, _), v in seen.items():
    v = v[-1]          # last repetition (warm)
    print('| %s | %.1f | %.1f | %.1f |' % (k, v[0], v[1], v[2]))
PY
done

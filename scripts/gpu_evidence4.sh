#!/bin/bash
# round 4, closing evidence on ONE box: (1) order-7 A/B per layer, (2) the choice applied to everything after it, (3) bench.py,
# (4) kernel traces (product and single-stream) + HBM traffic counters, (5) SQ counters of the dominant kernel
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
make -C oracle >/dev/null 2>&1
{ hostname; /opt/rocm/bin/rocm-smi --showuniqueid 2>/dev/null | grep -i "unique"; } > gpurun_out/ev4_box.txt 2>&1; cat gpurun_out/ev4_box.txt
C="per-tap:0:0:0,pp2:2:0:2,pp7:2:0:7"
B=16 LAYERS=conv8,conv13,conv18,conv20 CONFIGS=$C timeout 400 python scripts/pp_sweep.py > gpurun_out/pp7_b16.log 2>&1; cat gpurun_out/pp7_b16.log
# order 7 is taken when every result matches the per-tap kernels like order 2's does and the summed time is lower
CH=$(python - <<'PY'
import re
t2 = t7 = 0.0; ok = True
for l in open('gpurun_out/pp7_b16.log'):
    f = re.findall(r'([0-9.]+)\|\s*[0-9]+ \S+ ([0-9e.+-]+)', l)
    if len(f) == 3:
        t2 += float(f[1][0]); t7 += float(f[2][0])
        ok = ok and float(f[2][1]) <= max(2 * float(f[1][1]), 1e-2)
print(7 if ok and t7 > 0 and t7 < 0.985 * t2 else 2, t2, t7, ok)
PY
)
echo "choice: $CH" | tee gpurun_out/pp7_choice.txt
export YOLO2_PP_SCHED=${CH%% *}
python bench.py > gpurun_out/ev4_bench.log 2>&1; tail -1 gpurun_out/ev4_bench.log | cut -c1-600
bash scripts/gpu_traffic.sh > gpurun_out/ev4_traffic.log 2>&1; tail -5 gpurun_out/ev4_traffic.log
timeout 200 python -m pytest tests/test_kernels_gpu.py -k "tap_fused and (3x5x7 or 2x19x19 or 1x27x28 or 5x10x10)" -q -p no:cacheprovider --timeout 100 2>&1 | tail -3 > gpurun_out/ev4_tap_tests.log; cat gpurun_out/ev4_tap_tests.log
bash scripts/gpu_pmc.sh > gpurun_out/ev4_pmc.log 2>&1; tail -5 gpurun_out/ev4_pmc.log
python scripts/conv_bench.py "r04c" > gpurun_out/ev4_conv_bench.log 2>&1; tail -8 gpurun_out/ev4_conv_bench.log

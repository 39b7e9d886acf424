#!/bin/bash
# row-of-taps filter gradient: parity tests (toy shapes, every variant; bench shapes), the per-layer A/B against the per-tap kernel at batch 16 and 8,
# and the training step with and without it -- one box
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
{ hostname; /opt/rocm/bin/rocm-smi --showuniqueid 2>/dev/null | grep -i "unique"; } > gpurun_out/w3_box.txt 2>&1
timeout 300 python -m pytest tests/test_kernels_gpu.py -k "wgrad" -q -p no:cacheprovider -x 2>&1 | tail -25 > gpurun_out/w3_tests.log; tail -3 gpurun_out/w3_tests.log
timeout 600 python -m pytest tests/test_bench_shapes_gpu.py -k "wgrad" -q -p no:cacheprovider -x 2>&1 | tail -25 > gpurun_out/w3_bench_tests.log; tail -3 gpurun_out/w3_bench_tests.log
VARIANTS=2,0 timeout 300 python scripts/wgrad_ab.py ${1:-w3} > gpurun_out/w3_ab.log 2>&1; cat gpurun_out/w3_ab.log
B=8 VARIANTS=2,0 timeout 300 python scripts/wgrad_ab.py ${1:-w3}_b8 > gpurun_out/w3_ab_b8.log 2>&1; cat gpurun_out/w3_ab_b8.log
for i in 1 2; do for v in 2 0; do
  YOLO2_WGRAD_VARIANT=$v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-detect --no-kernel-timer 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('wgrad variant $v (2 = per-tap kernel everywhere, 0 = rule) run $i: %.3f ms/step %.0f img/s' % (j['ms_per_step'], j['value']))" | tee -a gpurun_out/w3_step.log
done; done

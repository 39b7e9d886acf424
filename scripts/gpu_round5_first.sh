#!/bin/bash
# First GPU call of the next round (everything here was written after round 4's GPU budget was spent; each step is bounded by its own timeout):
#  1. the tiny tap-fused shapes that hung before the stream-K grid clamp (tests/test_kernels_gpu.py, YOLO2_TEST_TINY_TAP_SHAPES)
#  2. in-kernel cycle stamps of the ping-pong kernel: cycles per phase and the shader clock a launch really gets (scripts/pp_phase_cycles.py)
#  3. the four-rows-in-flight BN consumers: gated tests, then the A/B per layer shape (scripts/bn_rows_in_flight.py)
#  4. the long-share launch-rule clause at batch 8 (YOLO2_PP_LONG_SHARE=26)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
make -C oracle >/dev/null 2>&1
{ hostname; /opt/rocm/bin/rocm-smi --showuniqueid 2>/dev/null | grep -i "unique"; } > gpurun_out/r5_box.txt 2>&1
YOLO2_TEST_TINY_TAP_SHAPES=1 timeout 240 python -m pytest tests/test_kernels_gpu.py -k "tap_fused and (3x5x7 or 2x19x19 or 1x27x28 or 5x10x10)" -q -p no:cacheprovider -x -o faulthandler_timeout=20 --durations=5 2>&1 | tail -25 > gpurun_out/r5_tiny_tap.log; tail -8 gpurun_out/r5_tiny_tap.log
# (build yolo_tf_amd/csrc/libyolo2hip_exp.so with scripts/pp_experiments_build.sh BEFORE the call: minutes of host time, and the .so travels)
EXP=$R/yolo_tf_amd/csrc/libyolo2hip_exp.so
if [ -f $EXP ]; then
  YOLO2_LIB_PATH=$EXP timeout 300 python scripts/pp_phase_cycles.py > gpurun_out/r5_phase_cycles.log 2>&1; cat gpurun_out/r5_phase_cycles.log
  YOLO2_LIB_PATH=$EXP GRID=2 LAYERS=conv8,conv5 timeout 200 python scripts/pp_phase_cycles.py > gpurun_out/r5_phase_cycles_tiles.log 2>&1; cat gpurun_out/r5_phase_cycles_tiles.log
else
  echo "no experiments library: run scripts/pp_experiments_build.sh first" | tee gpurun_out/r5_phase_cycles.log
fi
YOLO2_TEST_BN_ROWS4=1 timeout 200 python -m pytest tests/test_kernels_gpu.py -k "bn_consumers or rows_in_flight" -q -p no:cacheprovider -x 2>&1 | tail -6 > gpurun_out/r5_bn_rows4_tests.log; tail -3 gpurun_out/r5_bn_rows4_tests.log
timeout 200 python scripts/bn_rows_in_flight.py > gpurun_out/r5_bn_rows_in_flight.log 2>&1; cat gpurun_out/r5_bn_rows_in_flight.log
# batch-8 launch-rule candidate: COCO-80 batch 8 step with and without the long-share clause, same box, twice each
for i in 1 2; do
  for v in 0 26; do
    YOLO2_PP_LONG_SHARE=$v python bench.py --batch 8 --names 80 --steps 60 --warmup 10 --no-cpu-baseline --no-detect --no-kernel-timer 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('long_share $v run $i: %.3f ms/step %.0f img/s' % (j['ms_per_step'], j['value']))" | tee -a gpurun_out/r5_long_share_b8.log
  done
done

#!/bin/bash
# Round-6 kernels against the round-5 library on ONE box (profiles/r06_new_kernels.txt): the padded-index family (conv_c64.hip 128 / 32 filters,
# conv_c32.hip), the 1x1 data gradient with BN-backward sums (conv_d1.hip), the image layer's fused filter gradient (conv_first.hip).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash scripts/gpu_round6_kernels.sh'
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
OLD=$R/profiles/baseline/libyolo2hip_r05.so
BOX="box: hostname $(hostname), GPU $(/opt/rocm/bin/rocm-smi --showuniqueid 2>/dev/null | grep -i 'unique id: ' | head -1 | sed 's/.*Unique ID: *//'); commit $(cat .evidence_commit 2>/dev/null || echo '(snapshot)'); scripts/gpu_round6_kernels.sh (ONE gpurun call)"
{
echo "# round-6 kernels vs the round-5 library ($(cut -c1-7 profiles/baseline/libyolo2hip_r05.commit)), microbenchmarks (hipGraph replay of 20 launches / HIP events), batch 16 unless stated"
echo "# $BOX"
for which in r05 r06 r05 r06; do
  if [ $which = r05 ]; then export YOLO2_LIB_PATH=$OLD YOLO2_LIB_BASELINE=1; else unset YOLO2_LIB_PATH YOLO2_LIB_BASELINE; fi
  echo; echo "--- library $which: conv2 / conv4 forward, conv1 data gradient, conv1 forward (scripts/c64_bench.py)"
  BATCHES=16,64 timeout 300 python scripts/c64_bench.py 2>&1 | grep "^YOLO2_C64" | sed 's/^YOLO2_C64=1 //'
  echo "--- library $which: conv3 / conv6 data gradient + BN-backward sums (scripts/d1_bench.py)"
  timeout 300 python scripts/d1_bench.py 2>&1 | grep -v "Warn\|warn\|amdgpu.ids"
done
unset YOLO2_LIB_PATH YOLO2_LIB_BASELINE
echo; echo "--- image layer: BN / leaky / pool backward apply + filter gradient, two launches vs one (scripts/first_wgrad_bench.py; 512 partial rows here, <= 256 in a step)"
for i in 1 2; do timeout 300 python scripts/first_wgrad_bench.py 2>&1 | grep "^batch"; done
echo "--- ... grid sweep of the one-launch form (workgroups; default 512)"
for g in 256 512 768 1536; do echo "grid $g: $(YOLO2_FIRST_WGRAD_GRID=$g timeout 300 python scripts/first_wgrad_bench.py 2>&1 | grep '^batch' | sed 's/.*one launch//')"; done
echo; echo "--- training step, alternating (bench.py --steps 80): image-layer fusion off / on, conv_c64 off / on, conv_d1 off / on"
bash scripts/ab_step.sh 3 "first_two_launches=YOLO2_FUSE_FIRST_WGRAD=0" "c64_off=YOLO2_C64=0" "d1_off=YOLO2_D1=0" "c32_off=YOLO2_C32=0" "default="
} > gpurun_out/r06_new_kernels.txt 2>&1
tail -30 gpurun_out/r06_new_kernels.txt

#!/bin/bash
# Builds yolo_tf_amd/csrc/libyolo2hip_exp.so: the product library with the experiment instantiations of the kernels named on the command line
#   pp   conv_pp.hip      -DY2P_EXPERIMENTS   every SCHED variant, timing ablations, in-kernel cycle stamps   (scripts/pp_sweep.py, pp_phase_cycles.py)
#   w3   conv_wgrad3.hip  -DY2W3_EXPERIMENTS  variants 0 / 1 / 3, YOLO2_W3_ABL ablations and stamps           (scripts/wgrad_ab.py, w3_phase_cycles.py, gpu_w3_abl.sh)
#   c32  conv_c32.hip     -DY2C32_EXPERIMENTS YOLO2_C32_ABL ablations and stamps                               (scripts/c32_phase_cycles.py)
#   w32  conv_wgrad_c32.hip -DY2W32_EXPERIMENTS YOLO2_W32_ABL timing ablations                                 (CASES=16x208 scripts/w32_bench.py)
#   s4   conv_s4.hip      -DY2S_EXPERIMENTS   timing ablations and wall-clock phase sums (yolo2_debug_set_s4_abl)    (S4=1 scripts/pp_fixed_cost.py)
# usage: bash scripts/experiments_build.sh pp w3 c32      (default: all three)
# The product library is not touched; select the experiments build per process with YOLO2_LIB_PATH=$PWD/yolo_tf_amd/csrc/libyolo2hip_exp.so.
# Takes several minutes of host time with pp (every variant is a full instantiation of the kernel): build it HERE, before a gpurun call -- the
# .so travels with the snapshot -- not on the GPU box's clock.
cd "$(dirname "$0")/../yolo_tf_amd/csrc" || exit 1
python build.py > /dev/null || exit 1          # the product objects, up to date
WHAT="${*:-pp w3 c32 s4}"
HIPCC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result"
PP=conv_pp.o; W3=conv_wgrad3.o; C32=conv_c32.o; S4=conv_s4.o; D1=conv_d1.o; W32=conv_wgrad_c32.o
for w in $WHAT; do
  case $w in
    pp)  $HIPCC -DY2P_EXPERIMENTS -c conv_pp.hip -o conv_pp_exp.o || exit 1; PP=conv_pp_exp.o ;;
    w3)  $HIPCC -DY2W3_EXPERIMENTS -c conv_wgrad3.hip -o conv_wgrad3_exp.o || exit 1; W3=conv_wgrad3_exp.o ;;
    c32) $HIPCC -DY2C32_EXPERIMENTS -c conv_c32.hip -o conv_c32_exp.o || exit 1; C32=conv_c32_exp.o ;;
    s4)  $HIPCC -DY2S_EXPERIMENTS -c conv_s4.hip -o conv_s4_exp.o || exit 1; S4=conv_s4_exp.o ;;
    w32) $HIPCC -DY2W32_EXPERIMENTS -c conv_wgrad_c32.hip -o conv_wgrad_c32_exp.o || exit 1; W32=conv_wgrad_c32_exp.o ;;
    d1)  $HIPCC -DY2D1_EXPERIMENTS -c conv_d1.hip -o conv_d1_exp.o || exit 1; D1=conv_d1_exp.o ;;
    *) echo "unknown kernel '$w' (pp, w3, w32, c32, s4, d1)"; exit 1 ;;
  esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libyolo2hip_exp.so conv_igemm.o $PP $S4 conv_wgrad.o $W3 $W32 $C32 conv_c64.o $D1 conv_first.o elementwise.o head.o yolo1.o nms.o augment.o || exit 1
ls -la libyolo2hip_exp.so

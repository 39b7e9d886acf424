#!/bin/bash
# Timing-ablation builds of the implicit-GEMM kernels (experiments only, never the product): scripts/experiments/build/libabl_<bits>.so
# The hooks are NOT in the product source: scripts/experiments/conv_igemm_ablation_hooks.patch adds them to a copy of conv_igemm.hip
# (made against the revision it was cut from; re-cut it with `diff -u` after the kernel moves).
# usage: scripts/abl_build.sh 1 2 4 8 16 ...   (bit masks, Y2_ABL; ABLMACRO=Y2_TABL for the tap-fused kernel)
cd "$(dirname "$0")/.."; S=yolo_tf_amd/csrc; O=scripts/experiments/build; mkdir -p $O
cp $S/conv_igemm.hip $S/common.h $O/ && sed -i 's|"../../include/yolo2_hip.h"|"../../../include/yolo2_hip.h"|' $O/common.h
patch -s $O/conv_igemm.hip scripts/experiments/conv_igemm_ablation_hooks.patch || { echo "the ablation patch no longer applies: re-cut it"; exit 1; }
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result"
for b in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -D${ABLMACRO:-Y2_ABL}=$b -c $O/conv_igemm.hip -o $O/conv_igemm_$b.o &
done
wait
for b in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libabl_$b.so $O/conv_igemm_$b.o $S/conv_wgrad.o $S/conv_first.o $S/elementwise.o $S/head.o $S/yolo1.o $S/nms.o $S/augment.o
done
ls -la $O/*.so

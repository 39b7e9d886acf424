#!/bin/bash
# Timing-ablation builds of the implicit-GEMM kernel (experiments only): scripts/experiments/build/libabl_<bits>.so
# usage: scripts/abl_build.sh 1 2 4 8 16 ...   (bit masks, see Y2_ABL in csrc/conv_igemm.hip; ABLMACRO=Y2_TABL for the tap-fused kernel)
cd "$(dirname "$0")/.."; S=yolo_tf_amd/csrc; O=scripts/experiments/build; mkdir -p $O
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result"
for b in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -D${ABLMACRO:-Y2_ABL}=$b -c $S/conv_igemm.hip -o $O/conv_igemm_$b.o &
done
wait
for b in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libabl_$b.so $O/conv_igemm_$b.o $S/conv_wgrad.o $S/conv_first.o $S/elementwise.o $S/head.o $S/yolo1.o $S/nms.o $S/augment.o
done
ls -la $O/*.so

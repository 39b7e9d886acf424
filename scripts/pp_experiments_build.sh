#!/bin/bash
# Builds yolo_tf_amd/csrc/libyolo2hip_exp.so: the library with every SCHED variant, the timing-ablation instantiations and the in-kernel
# cycle stamps of conv_pp.hip (-DY2P_EXPERIMENTS).  The product library is not touched; select the experiments build per process with
#   YOLO2_LIB_PATH=$PWD/yolo_tf_amd/csrc/libyolo2hip_exp.so python scripts/pp_sweep.py        (CONFIGS=name:mode:grid:sched reaches the variants)
#   YOLO2_LIB_PATH=...                                     python scripts/pp_phase_cycles.py  (SCHED 2 + 512 + 4096)
# Takes several minutes of host time (every variant is a full instantiation of the kernel): build it HERE, before a gpurun call -- the
# .so travels with the snapshot -- not on the GPU box's clock.
cd "$(dirname "$0")/../yolo_tf_amd/csrc" || exit 1
python build.py > /dev/null || exit 1          # the other objects, up to date
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DY2P_EXPERIMENTS -c conv_pp.hip -o conv_pp_exp.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libyolo2hip_exp.so conv_igemm.o conv_pp_exp.o conv_wgrad.o conv_first.o elementwise.o head.o yolo1.o nms.o augment.o || exit 1
ls -la libyolo2hip_exp.so

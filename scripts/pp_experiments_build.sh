#!/bin/bash
# Rebuilds conv_pp.o with every SCHED variant and the timing-ablation instantiations (-DY2P_EXPERIMENTS), relinks libyolo2hip.so.
# scripts/pp_sweep.py CONFIGS=name:mode:grid:sched then reaches them; `python yolo_tf_amd/csrc/build.py --force` restores the product build.
# scripts/pp_phase_cycles.py: in-kernel cycle stamps per phase (SCHED 2 + 512 + 4096).
cd "$(dirname "$0")/../yolo_tf_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DY2P_EXPERIMENTS -c conv_pp.hip -o conv_pp.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libyolo2hip.so conv_igemm.o conv_pp.o conv_wgrad.o conv_first.o elementwise.o head.o yolo1.o nms.o augment.o

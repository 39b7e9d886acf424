#!/bin/bash
# run on the GPU box: per-kernel parity tests (all of them, no -x) + device info
mkdir -p gpurun_out
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/device.txt
make -C oracle >/dev/null 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/kernels.log
tail -40 gpurun_out/kernels.log

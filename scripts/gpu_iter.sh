#!/bin/bash
# scratch GPU iteration: targeted tests, then same-box A/B of the switches given as arguments
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "bn_consumers or maxpool" 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_network_gpu.py tests/test_reference_pins_gpu.py -x -q 2>&1 | tail -5
rm -f gpurun_out/bench_ab.txt
bash scripts/bench_ab.sh "$@"

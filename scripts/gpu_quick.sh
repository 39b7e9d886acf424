#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
make -C oracle >/dev/null 2>&1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -s 2>&1 | grep -E "bf16 cosine|^E  .*Assert|passed|failed|^FAILED" | cut -c1-300 > gpurun_out/pytest_gpu.log 2>&1
tail -8 gpurun_out/pytest_gpu.log
python scripts/conv_bench.py "current" > gpurun_out/conv_bench.log 2>&1; tail -16 gpurun_out/conv_bench.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-detect > gpurun_out/bench2.log 2>gpurun_out/bench2.err; tail -1 gpurun_out/bench2.log | cut -c1-300

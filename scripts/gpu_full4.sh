#!/bin/bash
# round 4: full GPU test-suite + the driver's bench command + per-layer table
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
make -C oracle >/dev/null 2>&1
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-detect > gpurun_out/full4_bench.log 2>gpurun_out/full4_bench.err; tail -1 gpurun_out/full4_bench.log | cut -c1-600
python scripts/conv_bench.py "r04" > gpurun_out/full4_conv_bench.log 2>&1; tail -18 gpurun_out/full4_conv_bench.log
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x 2>&1 | tail -30 > gpurun_out/full4_pytest.log; tail -12 gpurun_out/full4_pytest.log

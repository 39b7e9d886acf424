#!/bin/bash
# what the driver runs at round end: gpu tests, smoke(), default bench
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
make -C oracle >/dev/null 2>&1
timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/full_pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3 | tee gpurun_out/full_smoke.log
timeout 900 python bench.py > gpurun_out/full_bench.log 2> gpurun_out/full_bench.err; tail -1 gpurun_out/full_bench.log | cut -c1-400

#!/bin/bash
# One GPU call: the whole -m gpu suite (log with the PLAN lines), smoke, then the default bench line.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
T0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu -s --maxfail=40 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$? $(( $(date +%s) - T0 )) s"
grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -3
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu.log | head -40
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
T1=$(date +%s)
timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$? $(( $(date +%s) - T1 )) s"
tail -c 3000 gpurun_out/bench.json

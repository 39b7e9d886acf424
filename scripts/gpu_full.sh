#!/bin/bash
# run on the GPU box: all GPU tests, bench line, rocprofv3 kernel-trace summary
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
make -C $R/oracle >/dev/null 2>&1
cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -120 > gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps ${STEPS:-10} --warmup 3 ${BENCH_ARGS} > gpurun_out/bench.log 2> gpurun_out/bench.err
tail -3 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
if [ -z "$NO_PROF" ]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o run -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/prof.log 2>&1
  tail -3 $R/gpurun_out/prof.log
  find $R/gpurun_out/prof -name "*kernel_stats*" | head
  f=$(find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -40 "$f" > $R/gpurun_out/kernel_stats_head.csv
  # the raw trace is large: keep only the stats
  find $R/gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
fi

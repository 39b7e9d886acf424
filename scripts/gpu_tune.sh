#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
make -C oracle >/dev/null 2>&1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -s 2>&1 | grep -E "logits rel|^E  .*Assert|passed|failed|^FAILED" | cut -c1-330 > gpurun_out/pytest_gpu.log 2>&1
tail -12 gpurun_out/pytest_gpu.log
{
for cfg in "3 512" "3 0" "2 512" "4 512" "3 768"; do set -- $cfg
YOLO2_IGEMM_STAGES=$1 YOLO2_KSPLIT_BLOCKS=$2 python scripts/conv_bench.py "stages$1 split$2"
done
} > gpurun_out/conv_bench.log 2>&1
grep "totals" gpurun_out/conv_bench.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench2.log 2>gpurun_out/bench2.err; tail -1 gpurun_out/bench2.log | cut -c1-300

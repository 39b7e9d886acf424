#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
make -C oracle >/dev/null 2>&1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/pytest_gpu.log; tail -8 gpurun_out/pytest_gpu.log
{
YOLO2_KSPLIT_BLOCKS=0 YOLO2_IGEMM_CH=4 python scripts/conv_bench.py "ch4 nosplit"
YOLO2_KSPLIT_BLOCKS=0 YOLO2_IGEMM_CH=8 python scripts/conv_bench.py "ch8 nosplit"
YOLO2_KSPLIT_BLOCKS=512 YOLO2_IGEMM_CH=8 python scripts/conv_bench.py "ch8 split512"
YOLO2_KSPLIT_BLOCKS=768 YOLO2_IGEMM_CH=4 python scripts/conv_bench.py "ch4 split768"
YOLO2_KSPLIT_BLOCKS=512 YOLO2_IGEMM_CH=4 python scripts/conv_bench.py "ch4 split512"
YOLO2_KSPLIT_BLOCKS=1024 YOLO2_IGEMM_CH=4 python scripts/conv_bench.py "ch4 split1024"
} > gpurun_out/conv_bench.log 2>&1
grep "totals" gpurun_out/conv_bench.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench2.log 2>gpurun_out/bench2.err; tail -1 gpurun_out/bench2.log | cut -c1-400

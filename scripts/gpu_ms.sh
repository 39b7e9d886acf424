#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_network_gpu.py -q -x -k "multi_scale or train_step or full_size_training" -p no:cacheprovider 2>&1 | tail -8
timeout 300 python bench.py --multiscale --steps 20 --warmup 10 > gpurun_out/bench_ms.json 2> gpurun_out/bench_ms.err; tail -c 900 gpurun_out/bench_ms.json; tail -3 gpurun_out/bench_ms.err
timeout 300 python bench.py --batch 8 --names 80 --steps 30 --warmup 5 --no-cpu-baseline --no-detect > gpurun_out/bench_b8.json 2> gpurun_out/bench_b8.err; python -c "
import json; d=json.loads(open('gpurun_out/bench_b8.json').read().strip().splitlines()[-1]); print('batch 8 COCO-80:', d['value'], d['ms_per_step'], d['roofline']['frac'])"

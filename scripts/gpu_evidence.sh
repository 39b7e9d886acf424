#!/bin/bash
# ONE parametrised evidence script (replaces the per-experiment scripts/gpu_*.sh of rounds 1-4; they are in the git history).
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash scripts/gpu_evidence.sh r05 bench trace traffic pmc layers wgrad configs tests'
# Every section runs on the box of THIS call, writes gpurun_out/<tag>_<section>.* with the box id and commit in its first lines (copy what is to be
# judged into profiles/), and is bounded by its own timeout.  Sections:
#   bench    python bench.py (the driver's line: roofline, cpu_baseline, f32_parity_mode, detect)
#   trace    rocprofv3 --kernel-trace --stats of bench.py: product configuration + single stream, per-launch listing of the last step
#   traffic  the same + FETCH_SIZE / WRITE_SIZE in separate PMC passes; refreshes the dominant kernel's bytes per launch
#   pmc      SQ counters of forward + filter-gradient launches of six layers (scripts/one_layer.py), MFMA-busy calibrated on a pure MFMA loop
#   layers   per-layer microbenchmark table (scripts/conv_bench.py)
#   wgrad    filter gradient per layer, per-tap kernel vs the rule, batch 16 and 8, + the step A/B (scripts/gpu_w3.sh)
#   configs  the other BASELINE configurations' single-GPU legs and the batch fit (scripts/gpu_other_configs.sh)
#   tests    the whole -m gpu suite with durations
#   rounds   same-box A/B of this tree's library against the previous round's (PREV=r05: profiles/baseline/libyolo2hip_r05.so), alternating
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out; TAG=${1:-ev}; shift
make -C oracle >/dev/null 2>&1
BOX="box: hostname $(hostname), GPU $(/opt/rocm/bin/rocm-smi --showuniqueid 2>/dev/null | grep -i 'unique id: ' | head -1 | sed 's/.*Unique ID: *//'); commit $(cat .evidence_commit 2>/dev/null || echo '(snapshot)'); scripts/gpu_evidence.sh $TAG $* (ONE gpurun call)"
echo "# $BOX" | tee gpurun_out/${TAG}_box.txt
hdr() { echo "# $1"; echo "# $BOX"; echo; }
for sec in "$@"; do
  case $sec in
    bench)
      ( time timeout 600 python bench.py ) > gpurun_out/${TAG}_bench.log 2>&1
      { hdr "python bench.py (defaults: N = 1, 100 steps, 10 warm-up)"; grep '^{' gpurun_out/${TAG}_bench.log | tail -1; grep -E '^real' gpurun_out/${TAG}_bench.log; } > gpurun_out/${TAG}_bench_line.txt
      grep '^{' gpurun_out/${TAG}_bench.log | tail -1 | cut -c1-400 ;;
    trace|traffic)
      bash scripts/gpu_traffic.sh $([ $sec = trace ] && echo notraffic) > gpurun_out/${TAG}_traffic.log 2>&1
      for f in prof_summary.md prof1s_summary.md prof1s_last_step.txt prof_roofline.txt traffic_summary.md; do
        [ -f gpurun_out/$f ] && { hdr "$f of scripts/gpu_traffic.sh"; cat gpurun_out/$f; } > gpurun_out/${TAG}_$f
      done
      cat gpurun_out/prof_roofline.txt 2>/dev/null | tail -4 ;;
    pmc)
      bash scripts/gpu_pmc.sh > gpurun_out/${TAG}_pmc.log 2>&1
      { hdr "SQ counters (scripts/gpu_pmc.sh)"; cat gpurun_out/pmc/calibration.txt gpurun_out/pmc_summary.md; } > gpurun_out/${TAG}_sq_counters.md; tail -14 gpurun_out/pmc_summary.md | cut -c1-260 ;;
    layers)
      { hdr "per-layer microbenchmark, batch 16 bf16 (scripts/conv_bench.py: 20 launches per hipGraph replay)"; timeout 600 python scripts/conv_bench.py $TAG 2>&1 | grep -v amdgpu.ids; } > gpurun_out/${TAG}_conv_layers.txt; tail -3 gpurun_out/${TAG}_conv_layers.txt ;;
    wgrad)
      rm -f gpurun_out/w3_step.log; bash scripts/gpu_w3.sh $TAG > gpurun_out/${TAG}_w3.log 2>&1; grep "passed\|failed\|^network\|wgrad variant" gpurun_out/${TAG}_w3.log ;;
    configs)
      bash scripts/gpu_other_configs.sh > gpurun_out/${TAG}_configs.log 2>&1; { hdr "other BASELINE configurations, single-GPU legs (scripts/gpu_other_configs.sh)"; cat gpurun_out/other_configs.txt; } > gpurun_out/${TAG}_other_configs.txt; cat gpurun_out/other_configs.txt ;;
    tests)
      ( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 ) > gpurun_out/${TAG}_suite.log 2>&1
      { hdr "python -m pytest tests -m gpu"; tail -22 gpurun_out/${TAG}_suite.log; } > gpurun_out/${TAG}_gpu_tests.txt; tail -4 gpurun_out/${TAG}_suite.log ;;
    rounds)
      # THIS round's library against the previous round's (profiles/baseline/libyolo2hip_$PREV.so, scripts/build_baseline.sh) on THIS box, alternating:
      # the bench step, the per-layer table (the 24 dominant launches, the filter gradients).  Every "round N vs N-1" number of README / DESIGN comes from here.
      PREV=${PREV:-r05}; OLD=$R/profiles/baseline/libyolo2hip_$PREV.so
      { hdr "same-box A/B: this tree's library vs $PREV ($(cat profiles/baseline/libyolo2hip_$PREV.commit 2>/dev/null | cut -c1-7)), alternating runs of ONE call"
        if [ ! -f $OLD ]; then echo "no $OLD (bash scripts/build_baseline.sh $PREV <commit>)"; else
        for i in 1 2 3; do for which in old new; do
          if [ $which = old ]; then export YOLO2_LIB_PATH=$OLD YOLO2_LIB_BASELINE=1; else unset YOLO2_LIB_PATH YOLO2_LIB_BASELINE; fi
          timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-detect --no-f32 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j.get('roofline') or {}
        print('step  %-3s run $i: %.3f ms/step %6.0f img/s   dominant launches %.2f us = %.3f of peak' % ('$which', j['ms_per_step'], j['value'], 1e3*(r.get('avg_launch_ms') or 0), r.get('frac') or 0))"
        done; done
        for which in old new old new; do
          if [ $which = old ]; then export YOLO2_LIB_PATH=$OLD YOLO2_LIB_BASELINE=1; else unset YOLO2_LIB_PATH YOLO2_LIB_BASELINE; fi
          echo "--- per-layer table, $which library"; timeout 600 python scripts/conv_bench.py $which 2>&1 | grep -v amdgpu.ids
        done; unset YOLO2_LIB_PATH YOLO2_LIB_BASELINE; fi
      } > gpurun_out/${TAG}_rounds_ab.txt 2>&1; grep -E "^step|network-weighted|igemm launches" gpurun_out/${TAG}_rounds_ab.txt ;;
    *) echo "unknown section $sec" ;;
  esac
done

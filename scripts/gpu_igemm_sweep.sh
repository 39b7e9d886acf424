#!/bin/bash
mkdir -p gpurun_out; out=gpurun_out/igemm_sweep.txt; rm -f $out
for b in 16 8; do
for cfg in "X=0" "YOLO2_IGEMM_WIDE_NS2=0" "YOLO2_IGEMM_WIDE=0" "YOLO2_IGEMM_STREAM=0" "YOLO2_IGEMM_STREAM_MAXTILES=1024" "YOLO2_IGEMM_BM256=0" "YOLO2_IGEMM_TAP_MIN_STEPS=72" "YOLO2_IGEMM_TAP_MIN_STEPS=36 YOLO2_IGEMM_TAP_MIN_SHARE=20"; do
  echo "== B=$b $cfg" >> $out
  env B=$b LAYERS=conv2,conv5,conv8,conv13,conv18 $cfg timeout 300 python scripts/conv_bench.py sweep 2>/dev/null | grep "^conv" | awk '{print $1, $2, $3, $4, $5, $(NF-2), $(NF-1)}' >> $out
done; done
cat $out

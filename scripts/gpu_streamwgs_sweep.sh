#!/bin/bash
mkdir -p gpurun_out; out=gpurun_out/streamwgs_sweep.txt; rm -f $out
for b in 8 16; do
for w in 0 96 128 176 192 224; do
  echo "== B=$b YOLO2_IGEMM_STREAM_WGS=$w" >> $out
  env B=$b LAYERS=conv13,conv14,conv18,conv20 YOLO2_IGEMM_STREAM_WGS=$w timeout 300 python scripts/conv_bench.py sweep 2>/dev/null | grep "^conv" | awk '{print $1, $2, $3, $4, $5, $(NF-2), $(NF-1)}' >> $out
done; done
paste - - - - - < $out

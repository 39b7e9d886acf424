#!/bin/bash
mkdir -p gpurun_out; out=gpurun_out/wgrad_sweep2.txt; rm -f $out
for b in 16 8 32; do
for cfg in "YOLO2_WGRAD_BLOCKS=0" "YOLO2_WGRAD_BLOCKS=512 YOLO2_WGRAD_REMAP=0" "YOLO2_WGRAD_BLOCKS=360" "YOLO2_WGRAD_BLOCKS=432" "YOLO2_WGRAD_BLOCKS=432 YOLO2_WGRAD_REMAP=1" "YOLO2_WGRAD_BLOCKS=288" "YOLO2_WGRAD_BLOCKS=216"; do
  echo "== B=$b $cfg" >> $out
  env B=$b LAYERS=conv8 $cfg timeout 300 python scripts/conv_bench.py sweep 2>/dev/null | grep "^conv" | awk '{print $1, $6, $7, $NF}' >> $out
done; done
paste - - < $out

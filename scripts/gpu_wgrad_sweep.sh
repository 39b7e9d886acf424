#!/bin/bash
# filter-gradient block-count sweep on the layers whose grids are quantised badly (scripts/conv_bench.py, wgrad column)
mkdir -p gpurun_out; out=gpurun_out/wgrad_sweep.txt; rm -f $out
for t in 0 256 384 448 504 640 768 1024 1536 2048; do
  echo "== YOLO2_WGRAD_BLOCKS=$t" >> $out
  LAYERS=conv1,conv2,conv3,conv5,conv6,conv8,conv9,conv13,conv14 YOLO2_WGRAD_BLOCKS=$t timeout 300 python scripts/conv_bench.py sweep 2>/dev/null | grep "^conv" | awk '{print $1, $6, $7, $NF}' >> $out
done
cat $out

cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_kernels_gpu.py -k "conv_forward or conv_bn_fused or first or wgrad" -q -p no:cacheprovider -x 2>&1 | tail -4
timeout 300 python -m pytest tests/test_bench_shapes_gpu.py -k "conv0" -q -p no:cacheprovider -x 2>&1 | tail -3
timeout 300 python -m pytest tests/test_network_gpu.py -k "train_step_matches or full_size_training" -q -p no:cacheprovider -x 2>&1 | tail -3
for i in 1 2; do
  echo "-- before"; YOLO2_LIB_PATH=$PWD/yolo_tf_amd/csrc/libyolo2hip_before.so LAYERS=conv0 python scripts/conv_bench.py before 2>&1 | grep "^conv0"
  echo "-- after"; LAYERS=conv0 python scripts/conv_bench.py after 2>&1 | grep "^conv0"
done

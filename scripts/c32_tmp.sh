cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py tests/test_network_gpu.py tests/test_reference_pins_gpu.py -q -p no:cacheprovider -x 2>&1 | tail -4
for i in 1 2; do
  echo "-- before"; YOLO2_LIB_PATH=$PWD/yolo_tf_amd/csrc/libyolo2hip_before.so LAYERS=conv1 python scripts/conv_bench.py before 2>&1 | grep "^conv1" | cut -c1-50
  echo "-- after"; LAYERS=conv1 python scripts/conv_bench.py after 2>&1 | grep "^conv1" | cut -c1-50
done
for i in 1 2; do
YOLO2_LIB_PATH=$PWD/yolo_tf_amd/csrc/libyolo2hip_before.so python bench.py --steps 30 --warmup 5 --no-f32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('before', d['value'], d['ms_per_step'])"
python bench.py --steps 30 --warmup 5 --no-f32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('after ', d['value'], d['ms_per_step'])"
done

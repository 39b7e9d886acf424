#!/bin/bash
# the tests added since the last full GPU run (fast iteration)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_yolo1_gpu.py "tests/test_network_gpu.py::test_tensorflow_checkpoint_and_event_file_round_trip" tests/test_network_gpu.py -k "yolo1 or tensorflow or multi_scale" -x -q -s -m gpu 2>&1 | tail -40 | tee gpurun_out/new_tests.log

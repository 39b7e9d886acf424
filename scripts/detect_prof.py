"""Batch-256 detect loop (BASELINE configs[4]) for rocprofv3: forward (moving-stat BN) + decode + on-GPU NMS."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from yolo_tf_amd.session import DetectSession
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
b, _ = bench.make_builder('darknet', 20, 416, False, tempfile.mkdtemp())
sess = DetectSession(b, batch, dtype='bf16', seed=0)
images = torch.rand(batch, 416, 416, 3, device='cuda') * 255.0
for _ in range(3):
    sess.detect(images, 0.3, 0.4)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    sess.detect(images, 0.3, 0.4)
torch.cuda.synchronize()
print('detect batch %d: %.3f ms per batch' % (batch, (time.perf_counter() - t0) / 5 * 1e3))

"""Is the training loop host-bound?  Time to ENQUEUE K steps (Python + ctypes + HIP launch calls) vs time until the GPU has finished them.
    python scripts/host_bound.py [batch] [names]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from yolo_tf_amd.session import TrainSession
from yolo_tf_amd.utils import data
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
names = int(sys.argv[2]) if len(sys.argv) > 2 else 20
b, cfg = bench.make_builder('darknet', names, 416, True, tempfile.mkdtemp())
sess = TrainSession(b, B, dtype='bf16', optimizer='adam', learning_rate=1e-4, config=cfg, seed=0)
images = torch.rand(B, 416, 416, 3, device='cuda') * 255.0
sess.upload_labels(data.synthetic_batch(B, names, 13, 13, seed=1))
for _ in range(5):
    sess.step(images)
torch.cuda.synchronize()
for K in (20, 50):
    t0 = time.perf_counter()
    for _ in range(K):
        sess.step(images)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('B=%d K=%d: enqueue %.3f ms/step, until done %.3f ms/step (host %s the GPU)' % (B, K, (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3,
          'is AHEAD of' if (t1 - t0) < 0.9 * (t2 - t0) else 'is NOT ahead of'))

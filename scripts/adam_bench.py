"""Times yolo2_adam over the Darknet-19 arena size (67.16 M parameters, 1.88 GB of traffic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_tf_amd import ops
n = 67161104
w, g, m, v = [torch.randn(n, device='cuda') for _ in range(4)]
v.abs_()
for _ in range(3):
    ops.adam(w, g, m, v, n, 1e-3, 0.9, 0.999, 1e-8, 1.0)
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); a.record()
for _ in range(20):
    ops.adam(w, g, m, v, n, 1e-3, 0.9, 0.999, 1e-8, 1.0)
e.record(); torch.cuda.synchronize()
t = a.elapsed_time(e) / 20
print('adam: %.1f us, %.2f TB/s' % (t * 1e3, n * 28 / t / 1e9))

"""Does the RELATIVE placement of the four optimizer arenas change the Adam pass?  (HBM channel aliasing: element i of params, grads, m and v
is touched by the same lane at the same time.)  Arenas are carved from one allocation at controlled distances."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_tf_amd import ops
for n in (67161104, 67474304):
    for skew_kb in (0, 1, 4, 16, 64, 257, 1031):
        dist = ((n * 4 + (2 << 20) - 1) // (2 << 20)) * (2 << 20) + skew_kb * 1024        # bytes between arena starts: 2 MiB granules (what the caching allocator gives) + skew
        pool = torch.zeros((4 * dist) // 4 + 1024, device='cuda')
        ar = [pool[(k * dist) // 4:(k * dist) // 4 + n] for k in range(4)]
        ar[3].fill_(0.01)
        for _ in range(3):
            ops.adam(ar[0], ar[1], ar[2], ar[3], n, 1e-3, 0.9, 0.999, 1e-8, 1.0)
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(10):
            ops.adam(ar[0], ar[1], ar[2], ar[3], n, 1e-3, 0.9, 0.999, 1e-8, 1.0)
        e.record(); torch.cuda.synchronize()
        t = a.elapsed_time(e) / 10
        print('n %d  arena distance = 2 MiB granules + %4d KiB: %.1f us, %.2f TB/s' % (n, skew_kb, t * 1e3, n * 28 / t / 1e9))
        del pool, ar

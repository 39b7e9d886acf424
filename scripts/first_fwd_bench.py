"""conv0 forward (3 channels in an 8-wide pixel -> 32 filters at 416 x 416, conv_first.hip) with its batch statistics: us per launch, hipGraph replay over rotated
output buffers.  usage: [YOLO2_LIB_PATH=...] python scripts/first_fwd_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_tf_amd import ops
T = torch.bfloat16
ws = torch.zeros(1024 + 256 * 256 * 128, dtype=torch.float32, device='cuda')
for B in (16, 32):
    H = 416
    M = B * H * H
    x = torch.randn(M * 8, device='cuda').to(T)
    w = torch.randn(9 * 3 * 32, device='cuda') * 0.1
    F = torch.zeros(32 * 9 * 8, dtype=T, device='cuda')
    ops.filter_prep(w, F, None, 3, 3, 8, 32, 32, T)
    ys = [torch.zeros(M * 32, dtype=T, device='cuda') for _ in range(3)]
    part = torch.zeros(2 * 256 * 32, dtype=torch.float32, device='cuda')
    shift = torch.zeros(32, device='cuda')
    fns = [(lambda y=y: ops.conv2d_bn(x, F, y, ws, B, H, H, 8, 8, 32, 32, 3, shift, part)) for y in ys]
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    n = 12
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for i in range(n):
                fns[i % 3]()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n * 1e3)
    print('%s batch %d: conv0 forward + statistics %.1f us (%.2f TB/s of 80 B per pixel)  plan %s' % (
        os.path.basename(os.environ.get('YOLO2_LIB_PATH', 'product')), B, best, M * 80 / best * 1e-6, ops.last_conv_plan()), flush=True)

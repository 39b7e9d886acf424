"""conv2 / conv4 forward (64 -> 128 channels, 104 x 104) and conv1's data gradient (64 -> 32 at 208 x 208): conv_c64.hip against the generic per-tap
kernel, per epilogue, at batch 16 and 64 (a per-workgroup fixed cost -- the filter load into registers -- shows as a time that does not scale with
the batch).  Run in ONE process per library switch: YOLO2_C64=0 / 1 python scripts/c64_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_tf_amd import ops
T = torch.bfloat16
H, cin, cout = 104, 64, 128
ws = torch.zeros(1024 + 256 * 256 * 128, dtype=torch.float32, device='cuda')


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n * 1e3)
    return best


for B in [int(b) for b in os.environ.get("BATCHES", "16,64").split(",")]:
    M = B * H * H
    x = torch.randn(M * cin, device='cuda').to(T)
    y = torch.zeros(M * cout, dtype=T, device='cuda')
    w = torch.randn(9 * cin * cout, device='cuda') * 0.05
    F = torch.zeros(cout * 9 * cin, dtype=T, device='cuda')
    ops.filter_prep(w, F, None, 3, cin, cin, cout, cout, T)
    part = torch.zeros(2 * 256 * cout, dtype=torch.float32, device='cuda')
    shift = torch.zeros(cout, device='cuda')
    bias = torch.zeros(cout, device='cuda')
    res = []
    for name, fn in (('plain', lambda: ops.conv2d_ws(x, F, None, y, ws, B, H, H, cin, cin, cout, cout, 3)),
                     ('stats', lambda: ops.conv2d_bn(x, F, y, ws, B, H, H, cin, cin, cout, cout, 3, shift, part)),
                     ('bias+leaky', lambda: ops.conv2d_bias_leaky(x, F, bias, y, ws, B, H, H, cin, cin, cout, cout, 3, 0.1))):
        t = timed(fn)
        res.append('%s %.1f us (%.0f TFLOP/s)' % (name, t, 2.0 * M * cout * 9 * cin / t * 1e-6))
        plan = ops.last_conv_plan()
    print('YOLO2_C64=%s batch %d: %s   plan %s' % (os.environ.get('YOLO2_C64', '1'), B, ', '.join(res), '/'.join(str(plan[k]) for k in ('BM', 'BN', 'stages', 'grid_x'))))
    # conv1's data gradient: dY 64 channels at 208 x 208 -> dX 32 channels
    Hn, M = 208, B * 208 * 208
    dy = torch.randn(M * 64, device='cuda').to(T)
    dx = torch.zeros(M * 32, dtype=T, device='cuda')
    wn = torch.randn(9 * 32 * 64, device='cuda') * 0.05
    Fd = torch.zeros(32 * 9 * 64, dtype=T, device='cuda')
    ops.filter_prep(wn, None, Fd, 3, 32, 32, 64, 64, T)
    t = timed(lambda: ops.conv2d_ws(dy, Fd, None, dx, ws, B, Hn, Hn, 64, 64, 32, 32, 3))
    plan = ops.last_conv_plan()
    print('YOLO2_C64=%s batch %d: conv1 data gradient %.1f us (%.0f TFLOP/s, %.2f TB/s of 192 B per pixel)   plan %s' % (
        os.environ.get('YOLO2_C64', '1'), B, t, 2.0 * M * 32 * 9 * 64 / t * 1e-6, M * 192 / t * 1e-6, '/'.join(str(plan[k]) for k in ('BM', 'BN', 'stages', 'grid_x'))))
    # conv1 forward (conv_c32.hip): 32 -> 64 channels at 208 x 208
    x1 = torch.randn(M * 32, device='cuda').to(T)
    y1 = torch.zeros(M * 64, dtype=T, device='cuda')
    w1 = torch.randn(9 * 32 * 64, device='cuda') * 0.05
    F1 = torch.zeros(64 * 9 * 32, dtype=T, device='cuda')
    ops.filter_prep(w1, F1, None, 3, 32, 32, 64, 64, T)
    part1 = torch.zeros(2 * 256 * 64, dtype=torch.float32, device='cuda')
    shift1, bias1 = torch.zeros(64, device='cuda'), torch.zeros(64, device='cuda')
    res = []
    for name, fn in (('plain', lambda: ops.conv2d_ws(x1, F1, None, y1, ws, B, Hn, Hn, 32, 32, 64, 64, 3)),
                     ('stats', lambda: ops.conv2d_bn(x1, F1, y1, ws, B, Hn, Hn, 32, 32, 64, 64, 3, shift1, part1)),
                     ('bias+leaky', lambda: ops.conv2d_bias_leaky(x1, F1, bias1, y1, ws, B, Hn, Hn, 32, 32, 64, 64, 3, 0.1))):
        t = timed(fn)
        res.append('%s %.1f us (%.2f TB/s of 192 B per pixel)' % (name, t, M * 192 / t * 1e-6))
        plan = ops.last_conv_plan()
    print('YOLO2_C64=%s batch %d: conv1 forward %s   plan %s' % (os.environ.get('YOLO2_C64', '1'), B, ', '.join(res), '/'.join(str(plan[k]) for k in ('BM', 'BN', 'stages', 'grid_x'))))
    # conv2 / conv4's data gradient: dY 128 channels at 104 x 104 -> dX 64 channels
    Mw = B * 104 * 104
    dyw = torch.randn(Mw * 128, device='cuda').to(T)
    dxw = torch.zeros(Mw * 64, dtype=T, device='cuda')
    ww = torch.randn(9 * 64 * 128, device='cuda') * 0.05
    Fdw = torch.zeros(64 * 9 * 128, dtype=T, device='cuda')
    ops.filter_prep(ww, None, Fdw, 3, 64, 64, 128, 128, T)
    t = timed(lambda: ops.conv2d_ws(dyw, Fdw, None, dxw, ws, B, 104, 104, 128, 128, 64, 64, 3))
    plan = ops.last_conv_plan()
    print('YOLO2_C64=%s batch %d: conv2 / conv4 data gradient %.1f us (%.0f TFLOP/s)   plan %s' % (
        os.environ.get('YOLO2_C64', '1'), B, t, 2.0 * Mw * 64 * 9 * 128 / t * 1e-6, '/'.join(str(plan[k]) for k in ('BM', 'BN', 'stages', 'grid_x'))))

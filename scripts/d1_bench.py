"""1x1 data gradient + BN-backward sums of the wide early stages (conv3 at 104 x 104, conv6 at 52 x 52): conv_d1.hip against the generic kernel.
Run once per library switch: YOLO2_D1=0 / 1 python scripts/d1_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_tf_amd import ops
T = torch.bfloat16
B = int(os.environ.get('B', 16))
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


for name, H, k_ch, n_ch in (('conv3 dgrad (64 -> 128 at 104x104)', 104, 64, 128), ('conv6 dgrad (128 -> 256 at 52x52)', 52, 128, 256),
                            ('conv9 dgrad (256 -> 512 at 26x26: generic either way)', 26, 256, 512)):
    M = B * H * H
    dy = torch.randn(M * k_ch, device='cuda').to(T)
    dx = torch.zeros(M * n_ch, dtype=T, device='cuda')
    w = torch.randn(n_ch * k_ch, device='cuda') * 0.05
    F = torch.zeros(n_ch * k_ch, dtype=T, device='cuda')
    ops.filter_prep(w, None, F, 1, n_ch, n_ch, k_ch, k_ch, T)
    yprev = torch.randn(M * n_ch, device='cuda').to(T)
    mean, var = torch.zeros(n_ch, device='cuda'), torch.ones(n_ch, device='cuda')
    gamma, beta = torch.ones(n_ch, device='cuda'), torch.zeros(n_ch, device='cuda')
    dg, db = torch.zeros(n_ch, device='cuda'), torch.zeros(n_ch, device='cuda')
    part = torch.zeros(2 * 256 * n_ch, dtype=torch.float32, device='cuda')
    red = torch.zeros(ops.workspace_bytes('bn', n_ch) // 8, dtype=torch.float64, device='cuda')

    def fused():
        if ops.conv2d_dgrad_bn(dy, F, dx, ws, B, H, H, k_ch, k_ch, n_ch, n_ch, 1, yprev, mean, var, gamma, beta, dg, db, part, red, 1e-5, 0.1):
            ops.bn_part_to_grads(part, n_ch, dg, db)
    t_f = timed(fused)
    plan = ops.last_conv_plan()
    t_p = timed(lambda: ops.conv2d_ws(dy, F, None, dx, ws, B, H, H, k_ch, k_ch, n_ch, n_ch, 1))
    mb = M * (k_ch + 2 * n_ch) * 2 / 1e6
    print('YOLO2_D1=%s batch %d %s: dgrad + BN sums (+ part_to_grads) %.1f us = %.2f TB/s of its %.0f MB; plain dgrad %.1f us; plan %s' % (
        os.environ.get('YOLO2_D1', '1'), B, name, t_f, mb / t_f, mb, t_p, '/'.join(str(plan[k]) for k in ('BM', 'BN', 'stages', 'grid_x', 'grid_y'))))

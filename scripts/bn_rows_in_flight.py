"""A/B of the un-pooled BN consumers with one and with four pixel rows per thread in flight (yolo2_debug_set_bn_rows_in_flight; elementwise.hip
bn_leaky_fin4_kernel / bn_bwd_apply_fin4_kernel) on the Darknet-19 layer shapes, bf16, hipGraph-replayed; results compared bit for bit.
usage: [B=16] python scripts/bn_rows_in_flight.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_tf_amd import ops

B = int(os.environ.get('B', 16))
T = torch.bfloat16
SHAPES = [(104, 128, 1), (104, 64, 1), (52, 256, 1), (52, 128, 1), (26, 512, 2), (26, 256, 2), (13, 1024, 6), (13, 512, 2)]      # (H, C, layers of that shape per step)


def replay_us(fn, n=20):
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
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n * 1e3)
    return best


tot = {1: 0.0, 4: 0.0}
print('batch %d; us per launch (rows in flight 1 | 4), GB/s of the 4-row form, bit-identical?' % B)
for H, C, count in SHAPES:
    M = B * H * H
    rows = min(128, ops.bn_fin_rows_limit(C, T))
    y = (torch.randn(M * C, device='cuda') * 1.5).to(T)
    da = torch.randn(M * C, device='cuda').to(T)
    part = torch.rand(2 * 256 * C, device='cuda')
    shift = torch.randn(C, device='cuda') * 0.1
    gamma, beta = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.2
    mean_b, var_b = torch.randn(C, device='cuda') * 0.1, torch.rand(C, device='cuda') + 0.5
    mean, var = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    A = torch.zeros(M * C, dtype=T, device='cuda')
    dg, db = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    dY = torch.zeros(M * C, dtype=T, device='cuda')
    fwd = lambda: ops.bn_leaky_fin(y, part, rows, shift, mean, var, None, None, 0.999, gamma, beta, A, M, C, C, 1e-5, 0.1)
    bwd = lambda: ops.bn_leaky_bwd_apply_fin(da, C, y, mean_b, var_b, gamma, beta, part, rows, 256 * C, dg, db, dY, M, C, 1e-5, 0.1)
    t, out = {}, {}
    for n in (1, 4):
        ops.set_bn_rows_in_flight(n)
        t[n] = (replay_us(fwd), replay_us(bwd))
        out[n] = (A.clone(), dY.clone(), mean.clone(), dg.clone())
        tot[n] += count * (t[n][0] + t[n][1])
    same = all(torch.equal(a, b) for a, b in zip(out[1], out[4]))
    bytes_f, bytes_b = 2.0 * M * C * 2, 3.0 * M * C * 2
    print('%3dx%-3d C=%-4d x%d   apply %6.1f | %6.1f (%5.0f GB/s)   backward apply %6.1f | %6.1f (%5.0f GB/s)   %s' % (
        H, H, C, count, t[1][0], t[4][0], bytes_f / t[4][0] / 1e3, t[1][1], t[4][1], bytes_b / t[4][1] / 1e3, 'identical' if same else 'DIFFERENT'), flush=True)
ops.set_bn_rows_in_flight(1)
print('per step (layer counts applied): %.0f us -> %.0f us' % (tot[1], tot[4]))

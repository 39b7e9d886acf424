"""The image layer's filter gradient (conv_first.hip) at the bench shape (batch 16, 416 x 416, bf16): the two launches (BN / leaky / pool backward apply +
conv_first_wgrad_kernel) against the one launch that forms the output gradient in LDS (conv_first_wgrad_bn_kernel).
usage: [YOLO2_LIB_PATH=...] python scripts/first_wgrad_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_tf_amd import ops
T = torch.bfloat16
B, H, W = int(os.environ.get('B', 16)), 416, 416
M, MP = B * H * W, B * (H // 2) * (W // 2)
x = torch.randn(M * 8, device='cuda').to(T)
y = torch.randn(M * 32, device='cuda').to(T)
dp = torch.randn(MP * 32, device='cuda').to(T)
mean, var = torch.randn(32, device='cuda') * 0.1, torch.rand(32, device='cuda') + 0.5
gamma, beta = torch.rand(32, device='cuda') + 0.5, torch.randn(32, device='cuda') * 0.1
P0 = torch.zeros(MP * 32, dtype=T, device='cuda')
idx = torch.zeros(MP * 32, dtype=torch.uint8, device='cuda')
ops.bn_leaky_pool(y, mean, var, gamma, beta, P0, idx, B, H, W, 32, 32, 1e-5, 0.1)
ws = torch.zeros(ops.workspace_bytes('bn', 32) // 4 + 2 * 1024 * 32, dtype=torch.float32, device='cuda')
rows = ops.bn_leaky_pool_bwd_reduce_part(dp, 32, idx, y, mean, var, gamma, beta, ws, ops.bn_fin_rows_limit(32, T), B, H, W, 32, 1e-5, 0.1)
dg, db = torch.zeros(32, device='cuda'), torch.zeros(32, device='cuda')
dy = torch.zeros(M * 32, dtype=T, device='cuda')
dW = torch.zeros(9 * 3 * 32, dtype=torch.float32, device='cuda')


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


t_apply = timed(lambda: ops.bn_leaky_pool_bwd_apply_fin(dp, 32, idx, y, mean, var, gamma, beta, ws, rows, rows * 32, dg, db, dy, B, H, W, 32, 1e-5, 0.1))
t_wgrad = timed(lambda: ops.conv2d_wgrad(x, dy, dW, B, H, W, 3, 8, 32, 32, 3))
t_fused = timed(lambda: ops.first_layer_wgrad_bn(x, y, dp, 32, idx, mean, var, gamma, beta, ws, rows, rows * 32, dg, db, dW, B, H, W, 3, 1e-5, 0.1))
print('batch %d: apply_fin %.1f us + filter gradient %.1f us = %.1f us;  one launch %.1f us   (%d partial rows)' % (B, t_apply, t_wgrad, t_apply + t_wgrad, t_fused, rows))

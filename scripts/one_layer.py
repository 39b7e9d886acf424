"""Runs the forward conv of selected Darknet-19 layers a few times (for rocprofv3 counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_tf_amd import ops
LAYERS = {'conv1': (208, 32, 64, 3), 'conv2': (104, 64, 128, 3), 'conv5': (52, 128, 256, 3), 'conv8': (26, 256, 512, 3), 'conv13': (13, 512, 1024, 3), 'conv18': (13, 1024, 1024, 3), 'conv20': (13, 3072, 1024, 3)}
B, T = 16, torch.bfloat16
ws = torch.zeros(1024 + 256 * 256 * 128, dtype=torch.float32, device='cuda')
for name in sys.argv[1:]:
    H, cin, cout, k = LAYERS[name]
    M = B * H * H
    x = torch.randn(M * cin, device='cuda').to(T)
    y = torch.zeros(M * cout, dtype=T, device='cuda')
    dy = torch.randn(M * cout, device='cuda').to(T)
    w = torch.randn(k * k * cin * cout, device='cuda') * 0.05
    Ff = torch.zeros(cout * k * k * cin, dtype=T, device='cuda')
    dW = torch.zeros(k * k * cin * cout, dtype=torch.float32, device='cuda')
    ops.filter_prep(w, Ff, None, k, cin, cin, cout, cout, T)
    for _ in range(3):
        ops.conv2d_ws(x, Ff, None, y, ws, B, H, H, cin, cin, cout, cout, k)
        ops.conv2d_wgrad(x, dy, dW, B, H, H, cin, cin, cout, cout, k)
    torch.cuda.synchronize()

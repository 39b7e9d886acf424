"""Cycle stamps of conv1's filter-gradient kernel (conv_wgrad_c32.hip, experiments build: bash scripts/experiments_build.sh w32; YOLO2_W32_ABL=512).
usage: YOLO2_LIB_PATH=$PWD/yolo_tf_amd/csrc/libyolo2hip_exp.so YOLO2_W32_ABL=512 python scripts/w32_phase_cycles.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yolo_tf_amd import ops
T = torch.bfloat16
B, H = int(os.environ.get('B', 16)), 208
M = B * H * H
x = torch.randn(M * 32, device='cuda').to(T)
dy = torch.randn(M * 64, device='cuda').to(T)
blocks = 256
dW = torch.zeros(18432 + blocks * 12 * 16, dtype=torch.float32, device='cuda')
for _ in range(3):
    dW.zero_()
    ops.conv2d_wgrad(x, dy, dW, B, H, H, 32, 32, 64, 64, 3)
torch.cuda.synchronize()
plan = ops.last_wgrad_plan()
n = plan['blocks']
st = dW[18432:18432 + n * 12 * 16].cpu().numpy().view(np.uint64).reshape(n, 12, 8).astype(np.float64)
names = ['kernel', 'prologue', 'row loop', 'drain+barrier', 'LDS reduction', 'atomics issue', 'rows', 'atomics acknowledged']
print('conv1 filter gradient, batch %d, %d workgroups x 12 waves; 100 MHz ticks -> us' % (B, n))
for grp, sel in (('pixel group 0 (the three waves that write)', st[:, :3, :]), ('pixel groups 1-3', st[:, 3:, :])):
    print(grp)
    for i, nm in enumerate(names):
        if nm == 'rows':
            continue
        v = sel[:, :, i].ravel() / 100.0
        print('  %-22s mean %7.2f  min %7.2f  max %7.2f us' % (nm, v.mean(), v.min(), v.max()))

"""Where a workgroup of the 32 -> 64 channel forward kernel (conv_c32.hip) spends its time: s_memtime ticks per phase and tile of wave 0.
Build the experiments library first (bash scripts/experiments_build.sh c32), then:
  YOLO2_LIB_PATH=.../libyolo2hip_exp.so YOLO2_C32_ABL=8 python scripts/c32_phase_cycles.py
(YOLO2_C32_ABL bits 1 / 2 / 4 = no MFMA loop / no stores / no DMA after the first tile: timing ablations, wrong results)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_tf_amd import ops, _lib
B, H, cin, cout, T = 16, 208, 32, 64, torch.bfloat16
M = B * H * H
ws = torch.zeros(1024 + 256 * 256 * 128, dtype=torch.float32, device='cuda')
x = torch.randn(M * cin, device='cuda').to(T); y = torch.zeros(M * cout, dtype=T, device='cuda')
w = torch.randn(9 * cin * cout, device='cuda') * 0.05
Ff = torch.zeros(cout * 9 * cin, dtype=T, device='cuda')
ops.filter_prep(w, Ff, None, 3, cin, cin, cout, cout, T)
part = torch.zeros(2 * 256 * cout, dtype=torch.float32, device='cuda')
shift = torch.zeros(cout, device='cuda')
STATS = os.environ.get('STATS', '0') == '1'      # the training forward (MODE 1: + batch-norm partial sums)
for _ in range(3):
    if STATS:
        ops.conv2d_bn(x, Ff, y, ws, B, H, H, cin, cin, cout, cout, 3, shift, part)
    else:
        ops.conv2d_ws(x, Ff, None, y, ws, B, H, H, cin, cin, cout, cout, 3)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ['YOLO2_LIB_PATH'])
buf = (ctypes.c_ulonglong * (256 * 8))()
print('rc', lib.yolo2_debug_c32_stamps(buf))
d = torch.tensor(list(buf), dtype=torch.float64).view(256, 8)
names = ['barrier', 'dma issue', 'mfma loop', 'vmcnt wait', 'epilogue + stores', 'final sums (per tile)', 'total', 'tiles']
print('stats' if STATS else 'plain')
for sel, lab in ((d[:, 7] == 6, '6-tile WGs'), (d[:, 7] == 5, '5-tile WGs'), (d[:, 7] == 4, '4-tile WGs')):
    if int(sel.sum()) == 0:
        continue
    m = d[sel].mean(0)
    print(lab, int(sel.sum()), ' | '.join('%s %.0f' % (n, v / (m[7] if i < 6 else 1)) for i, (n, v) in enumerate(zip(names, m))))

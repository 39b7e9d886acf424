"""Where a workgroup of the row-of-taps filter-gradient kernel spends its shader cycles (conv_wgrad3.hip built with -DY2W3_EXPERIMENTS: bash scripts/experiments_build.sh w3, YOLO2_W3_ABL=512:
s_memtime at the phase boundaries of every super-step).  usage: YOLO2_LIB_PATH=.../libyolo2hip_exp.so YOLO2_W3_ABL=512 [VARIANTS=12,13,14] python scripts/w3_phase_cycles.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_tf_amd import ops

LAYERS = [('conv8', 26, 256, 512), ('conv13', 13, 512, 1024), ('conv18', 13, 1024, 1024), ('conv20', 13, 3072, 1024)]
if os.environ.get('LAYERS'):
    LAYERS = [l for l in LAYERS if l[0] in os.environ['LAYERS'].split(',')]
B, T = 16, torch.bfloat16
for name, H, cin, cout in LAYERS:
    M = B * H * H
    x = torch.randn(M * cin, device='cuda').to(T)
    dy = torch.randn(M * cout, device='cuda').to(T)
    for v in [int(a) for a in os.environ.get('VARIANTS', '12,13,14').split(',')]:
        ops.set_wgrad_variant(v)
        dW = torch.zeros(9 * cin * cout + 2 * 8 * 8 * 8192, dtype=torch.float32, device='cuda')
        for _ in range(2):
            ops.conv2d_wgrad(x, dy, dW, B, H, H, cin, cin, cout, cout, 3)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.conv2d_wgrad(x, dy, dW, B, H, H, cin, cin, cout, cout, 3)
        b.record()
        torch.cuda.synchronize()
        plan = ops.last_wgrad_plan()
        nw = plan['waves']
        d = dW[9 * cin * cout:].view(torch.int64)[:plan['blocks'] * nw * 8].view(-1, 8).double()
        d = d[d[:, 7] > 0]
        steps = d[:, 7].mean().item()
        m = d.mean(0)
        print('%-7s variant %d  blocks %4d waves %d  launch %.1f us | per wave: kernel %.0f cycles (max %.0f), prologue %.0f, epilogue %.0f; per super-step (%.1f steps): '
              'wait+barrier %.0f  DMA issue %.0f  reads+MFMA issue %.0f  table %.0f  (sum %.0f)' % (
                  name, v - 10, plan['blocks'], nw, a.elapsed_time(b) * 1e3, m[0], d[:, 0].max().item(), m[5], m[6], steps, m[1] / steps, m[2] / steps, m[3] / steps, m[4] / steps,
                  (m[1] + m[2] + m[3] + m[4]) / steps), flush=True)
ops.set_wgrad_variant(0)

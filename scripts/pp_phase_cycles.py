"""Where a K step of the ping-pong 3x3 kernel spends its shader cycles, measured inside the kernel (conv_pp.hip SCHED +4096 +512: s_memtime at the
phase boundaries of every step, the epilogue's stores replaced by each wave's sums), and the shader clock the launch actually ran at
(kernel cycles / kernel duration).  Needs the experiments library: bash scripts/experiments_build.sh pp (on the host, before the gpurun call)
usage: YOLO2_LIB_PATH=$PWD/yolo_tf_amd/csrc/libyolo2hip_exp.so [B=16] [LAYERS=conv8,conv20] [GRID=0|1|2] python scripts/pp_phase_cycles.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_tf_amd import ops

LAYERS = [('conv5', 52, 128, 256), ('conv8', 26, 256, 512), ('conv13', 13, 512, 1024), ('conv18', 13, 1024, 1024), ('conv20', 13, 3072, 1024)]
if os.environ.get('LAYERS'):
    LAYERS = [l for l in LAYERS if l[0] in os.environ['LAYERS'].split(',')]
B = int(os.environ.get('B', 16))
GRID = int(os.environ.get('GRID', 0))
T = torch.bfloat16
SCHED_TIME, SCHED_NOEPI = 2 + 512 + 4096 + int(os.environ.get('ABL', 0)), 2 + 512      # ABL: +256 no DMA in the loop, +128 no fragment reads, +64 no MFMAs (stamped launch only)
ws = torch.zeros(1024 + 256 * 256 * 128, dtype=torch.float32, device='cuda')


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


print('batch %d, grid mode %d; cycles per K step and wave (mean over the waves of a group); clock = kernel cycles / duration of the stamped launch' % (B, GRID))
for name, H, cin, cout in LAYERS:
    M = B * H * H
    x = torch.randn(M * cin, device='cuda').to(T)
    y = torch.zeros(M * cout, dtype=T, device='cuda')
    w = torch.randn(9 * cin * cout, device='cuda') * 0.05
    Ff = torch.zeros(cout * 9 * cin, dtype=T, device='cuda')
    ops.filter_prep(w, Ff, None, 3, cin, cin, cout, cout, T)
    fn = lambda: ops.conv2d_ws(x, Ff, None, y, ws, B, H, H, cin, cin, cout, cout, 3)
    ops.set_igemm_tap(2)
    res = {}
    for label, sched in (('product', 2), ('no-epilogue', SCHED_NOEPI), ('stamped', SCHED_TIME)):
        ops.set_pp(grid=GRID, dmapos=sched, min_steps=0, min_share=0)
        y.zero_()
        res[label] = replay_us(fn)
        plan = ops.last_conv_plan()
    torch.cuda.synchronize()
    G = plan['grid_x']
    d = y.view(torch.int64)[:G * 8 * 8].reshape(G * 8, 8).cpu().double()
    ok = d[:, 6] > 0
    per = d[ok, 1:6] / d[ok, 6:7]
    grp = (d[ok, 7].long() % 8) >= 4
    kcyc = d[:, 0]
    clock = kcyc.max().item() / res['stamped'] / 1e3      # cycles per us -> GHz
    print('%-7s grid %3d   us/launch: product %.1f, without epilogue %.1f, stamped %.1f   shader clock %.2f GHz (longest workgroup %.0f cycles)' % (
        name, G, res['product'], res['no-epilogue'], res['stamped'], clock, kcyc.max().item()))
    for gname, sel in (('waves 0-3', ~grp), ('waves 4-7', grp)):
        m = per[sel].mean(0)
        print('        %s: LOAD issue %4.0f  LOAD wait %4.0f  barrier %4.0f | MFMA issue %4.0f  barrier %4.0f | step %4.0f cycles  (steps per wave %.1f)' % (
            gname, m[0], m[1], m[2], m[3], m[4], m.sum(), d[ok, 6][sel].mean()))
ops.set_pp(grid=0, dmapos=2, min_steps=18, min_share=24)

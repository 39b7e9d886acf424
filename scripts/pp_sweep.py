"""Per-layer A/B of the 3x3 implicit-GEMM variants on the Darknet-19 shapes (bf16): per-tap kernels (mode 0), the round-2 tap-fused kernel
(mode 1: until commit 2e72607+3; gone since) and the ping-pong kernel (mode 2) as stream-K / one workgroup per tile with its SCHED variants
(conv_pp.hip).  Every configuration's outputs are also compared with the per-tap kernels' on the same operands (max |diff| / max |ref|).
usage: [YOLO2_LIB_PATH=.../libyolo2hip_exp.so for SCHED variants other than 2: scripts/experiments_build.sh pp] B=16 [LAYERS=conv8,conv20] [CONFIGS=name:mode:grid:sched,...] python scripts/pp_sweep.py   -> us | TFLOP/s per launch, one box, hipGraph-replayed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_tf_amd import ops

LAYERS = [('conv5', 52, 128, 256), ('conv8', 26, 256, 512), ('conv13', 13, 512, 1024), ('conv18', 13, 1024, 1024), ('conv20', 13, 3072, 1024)]
if os.environ.get('LAYERS'):
    LAYERS = [l for l in LAYERS if l[0] in os.environ['LAYERS'].split(',')]
B = int(os.environ.get('B', 16))
SIZE = int(os.environ.get('SIZE', 416))      # network input size: the layers' image sizes scale with it (multi-scale training: 320 .. 608)
LAYERS = [(l[0], l[1] * SIZE // 416) + tuple(l[2:]) for l in LAYERS]
T = torch.bfloat16
CONFIGS = [('per-tap', 0, 0, 0), ('RULE', 2, 0, 2), ('pp-sk', 2, 1, 2), ('pp-tile', 2, 2, 2)]      # RULE = the product launch rule (its thresholds); the others force the kernel wherever it can run
if os.environ.get('CONFIGS'):
    CONFIGS = [(c.split(':')[0],) + tuple(int(v) for v in c.split(':')[1:]) for c in os.environ['CONFIGS'].split(',')]
WHAT = os.environ.get('WHAT', 'fwd+stats,dgrad,dgrad+bn').split(',')


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(20):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 20 * 1e3)
    return best


ws = torch.zeros(1024 + 256 * 256 * 128, dtype=torch.float32, device='cuda')
print('batch %d; columns: %s   (us|TFLOP/s plan BM/stages/grid, rel. difference to the first column)' % (B, ', '.join('%s[%d/%d/%d]' % c for c in CONFIGS)))
for name, H, cin, cout in LAYERS:
    M = B * H * H
    x = torch.randn(M * cin, device='cuda').to(T)
    dy = torch.randn(M * cout, device='cuda').to(T)
    yprev = torch.randn(M * cin, device='cuda').to(T)
    y = torch.zeros(M * cout, dtype=T, device='cuda')
    dx = torch.zeros(M * cin, dtype=T, device='cuda')
    w = torch.randn(9 * cin * cout, device='cuda') * 0.05
    Ff = torch.zeros(cout * 9 * cin, dtype=T, device='cuda')
    Fd = torch.zeros(cin * 9 * cout, dtype=T, device='cuda')
    part = torch.zeros(2 * 256 * max(cin, cout), dtype=torch.float32, device='cuda')
    shift = torch.zeros(cout, dtype=torch.float32, device='cuda')
    pm, pv = torch.zeros(cin, device='cuda'), torch.ones(cin, device='cuda')
    pg, pb = torch.ones(cin, device='cuda'), torch.zeros(cin, device='cuda')
    dg, db = torch.zeros(cin, device='cuda'), torch.zeros(cin, device='cuda')
    red = torch.zeros(ops.workspace_bytes('bn', cin) // 8, dtype=torch.float64, device='cuda')
    ops.filter_prep(w, Ff, Fd, 3, cin, cin, cout, cout, T)
    fl = 2.0 * M * 9 * cin * cout
    calls = {'fwd+stats': (lambda: ops.conv2d_bn(x, Ff, y, ws, B, H, H, cin, cin, cout, cout, 3, shift, part), y),
             'dgrad': (lambda: ops.conv2d_ws(dy, Fd, None, dx, ws, B, H, H, cout, cout, cin, cin, 3), dx),
             'dgrad+bn': (lambda: ops.conv2d_dgrad_bn(dy, Fd, dx, ws, B, H, H, cout, cout, cin, cin, 3, yprev, pm, pv, pg, pb, dg, db, part, red, 1e-3, 0.1), dx)}
    rows = {k: [] for k in WHAT}
    ref = {}
    for cname, mode, grid, sched in CONFIGS:
        ops.set_igemm_tap(mode)
        if cname == 'RULE':
            ops.set_pp(grid=0, dmapos=sched, min_steps=18, min_share=24)
        else:
            ops.set_pp(grid=grid, dmapos=sched, min_steps=0, min_share=0)
        for what in WHAT:
            fn, out = calls[what]
            try:
                if (sched & 32) and sched >= 0 and what != 'dgrad':
                    rows[what].append('-')
                    continue
                out.zero_()
                t = timeit(fn)
                p = ops.last_conv_plan()
                o = out.float()
                if what not in ref:
                    ref[what] = o
                    diff = 0.0
                else:
                    diff = float((o - ref[what]).abs().max() / ref[what].abs().max())
                rows[what].append('%6.1f|%4.0f %d/%d/%d %.0e' % (t, fl / t / 1e6, p['BM'], p['stages'], p['grid_x'], diff))
            except Exception as e:      # noqa
                rows[what].append('ERR %s' % str(e)[:40])
    for what in rows:
        print('%-7s %-9s %s' % (name, what, '   '.join(rows[what])), flush=True)
ops.set_igemm_tap(2)
ops.set_pp(grid=0, dmapos=-1, min_steps=18, min_share=24)

"""Per-layer microbenchmark of the conv kernels on the Darknet-19 shapes (bf16).
usage: python scripts/conv_bench.py [tag]   env: B (batch, default 16); library knobs (YOLO2_*) are read by the library.
Prints time | TFLOP/s per layer for forward, data gradient and filter gradient and the variant each launch took."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_tf_amd import ops

LAYERS = [  # name, H, Cin, Cout, k, launches of this shape per network pass
    ('conv0', 416, 3, 32, 3, 1), ('conv1', 208, 32, 64, 3, 1), ('conv2', 104, 64, 128, 3, 2), ('conv3', 104, 128, 64, 1, 1),
    ('conv5', 52, 128, 256, 3, 2), ('conv6', 52, 256, 128, 1, 1), ('conv8', 26, 256, 512, 3, 3), ('conv9', 26, 512, 256, 1, 2),
    ('conv13', 13, 512, 1024, 3, 3), ('conv14', 13, 1024, 512, 1, 2), ('conv18', 13, 1024, 1024, 3, 2), ('conv20', 13, 3072, 1024, 3, 1),
    ('convout', 13, 1024, 125, 1, 1)]
B = int(os.environ.get('B', 16))
ONLY = os.environ.get('LAYERS')
T = torch.bfloat16
tag = sys.argv[1] if len(sys.argv) > 1 else ''


GRAPH = os.environ.get('GRAPH', '1') != '0'
if os.environ.get('PP_SCHED'):       # (A/B of a conv_pp.hip SCHED variant over the whole table)
    ops.set_pp(dmapos=int(os.environ['PP_SCHED']))


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if GRAPH:
        # 20 launches captured in one graph: the replay is not bounded by the ~10 us per launch the Python/ctypes host path costs
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
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / 60 * 1e3
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3   # us


def short(plan, keys):
    return '/'.join(str(plan[k]) for k in keys)


ws = torch.zeros(1024 + 256 * 256 * 128, dtype=torch.float32, device='cuda')
tot = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
f3 = {'fwd': [0.0, 0.0], 'dgrad': [0.0, 0.0]}      # [flops, us] of the 3x3 launches with > 64 filters (the roofline kernel set)
print('%-8s %12s %12s %12s   (us | TFLOP/s)  batch %d  %s' % ('layer', 'fwd', 'dgrad', 'wgrad', B, tag))
for name, H, cin, cout, k, mult in LAYERS:
    if ONLY and name not in ONLY.split(','):
        continue
    ldx, ldy = ops.pad8(cin), ops.pad8(cout)
    M = B * H * H
    x = torch.randn(M * ldx, device='cuda').to(T)
    dy = torch.randn(M * ldy, device='cuda').to(T)
    y = torch.zeros(M * ldy, dtype=T, device='cuda')
    dx = torch.zeros(M * ldx, dtype=T, device='cuda')
    w = torch.randn(k * k * cin * cout, device='cuda') * 0.05
    Ff = torch.zeros(cout * k * k * ldx, dtype=T, device='cuda')
    Fd = torch.zeros(cin * k * k * ldy, dtype=T, device='cuda')
    dW = torch.zeros(k * k * cin * cout, dtype=torch.float32, device='cuda')
    part = torch.zeros(2 * 256 * ldy, dtype=torch.float32, device='cuda')
    shift = torch.zeros(ldy, dtype=torch.float32, device='cuda')
    ops.filter_prep(w, Ff, Fd, k, cin, ldx, cout, ldy, T)
    fl = 2.0 * M * k * k * cin * cout
    bn = name != 'convout' and os.environ.get('BENCH_BN', '1') != '0'

    def fwd():
        if bn:      # as the engine's training forward: statistics from the epilogue (the partial rows are simply left to accumulate)
            ops.conv2d_bn(x, Ff, y, ws, B, H, H, ldx, ldx, cout, ldy, k, shift, part)
        else:
            ops.conv2d_ws(x, Ff, None, y, ws, B, H, H, ldx, ldx, cout, ldy, k)
    t_f = timeit(fwd)
    pf = short(ops.last_conv_plan(), ('BM', 'BN', 'chunks', 'stages', 'split'))
    t_d, pd = 0.0, '-'
    if name != 'conv0':
        t_d = timeit(lambda: ops.conv2d_ws(dy, Fd, None, dx, ws, B, H, H, ldy, ldy, cin, ldx, k))
        pd = short(ops.last_conv_plan(), ('BM', 'BN', 'chunks', 'stages', 'split'))
    t_w = timeit(lambda: ops.conv2d_wgrad(x, dy, dW, B, H, H, cin, ldx, cout, ldy, k))
    pw = short(ops.last_wgrad_plan(), ('BC', 'BN', 'pair', 'ranges', 'direct'))
    tot['fwd'] += t_f * mult; tot['dgrad'] += t_d * mult; tot['wgrad'] += t_w * mult
    if k == 3 and cout > 64:
        f3['fwd'][0] += fl * mult; f3['fwd'][1] += t_f * mult
    if k == 3 and cin > 64 and name != 'conv0':
        f3['dgrad'][0] += fl * mult; f3['dgrad'][1] += t_d * mult
    f = lambda t: '%7.1f|%4.0f' % (t, fl / t / 1e6) if t > 0 else '      -     '
    print('%-8s %s %s %s   %s  %s  %s' % (name, f(t_f), f(t_d), f(t_w), pf, pd, pw))
print('network-weighted totals (us): fwd %.0f dgrad %.0f wgrad %.0f  sum %.0f   %s' % (tot['fwd'], tot['dgrad'], tot['wgrad'], sum(tot.values()), tag))
fl3 = f3['fwd'][0] + f3['dgrad'][0]
us3 = f3['fwd'][1] + f3['dgrad'][1]
if us3 > 0:
    print('3x3 igemm launches with > 64 filters: %.1f TFLOP/s = %.3f of 2.5 PF (fwd %.0f, dgrad %.0f); wgrad network-weighted %.0f TFLOP/s   %s'
          % (fl3 / us3 / 1e6, fl3 / us3 / 1e6 / 2500, f3['fwd'][0] / max(f3['fwd'][1], 1e-9) / 1e6, f3['dgrad'][0] / max(f3['dgrad'][1], 1e-9) / 1e6,
             34.898e9 * B / max(tot['wgrad'], 1e-9) / 1e6, tag))

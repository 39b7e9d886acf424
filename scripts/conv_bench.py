"""Per-layer microbenchmark of the conv kernels on the Darknet-19 shapes (batch 16, bf16).
usage: python scripts/conv_bench.py [tag]   (env knobs YOLO2_IGEMM_CH / YOLO2_KSPLIT_BLOCKS are read by the library)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_tf_amd import ops

LAYERS = [  # name, H, Cin, Cout, k
    ('conv0', 416, 3, 32, 3), ('conv1', 208, 32, 64, 3), ('conv2', 104, 64, 128, 3), ('conv3', 104, 128, 64, 1),
    ('conv5', 52, 128, 256, 3), ('conv6', 52, 256, 128, 1), ('conv8', 26, 256, 512, 3), ('conv9', 26, 512, 256, 1),
    ('conv13', 13, 512, 1024, 3), ('conv14', 13, 1024, 512, 1), ('conv18', 13, 1024, 1024, 3), ('conv20', 13, 3072, 1024, 3),
    ('convout', 13, 1024, 125, 1)]
B = int(os.environ.get('B', 16))
T = torch.bfloat16
tag = sys.argv[1] if len(sys.argv) > 1 else ''


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3   # us


ws = torch.zeros(1024 + 256 * 256 * 128, dtype=torch.float32, device='cuda')
tot = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
print('%-8s %12s %12s %12s   (us | TFLOP/s)  %s' % ('layer', 'fwd', 'dgrad', 'wgrad', tag))
for name, H, cin, cout, k in LAYERS:
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
    ops.filter_prep(w, Ff, Fd, k, cin, ldx, cout, ldy, T)
    fl = 2.0 * M * k * k * cin * cout
    t_f = timeit(lambda: ops.conv2d_ws(x, Ff, None, y, ws, B, H, H, ldx, ldx, cout, ldy, k))
    t_d = timeit(lambda: ops.conv2d_ws(dy, Fd, None, dx, ws, B, H, H, ldy, ldy, cin, ldx, k)) if name != 'conv0' else 0.0
    t_w = timeit(lambda: ops.conv2d_wgrad(x, dy, dW, B, H, H, cin, ldx, cout, ldy, k))
    mult = {'conv2': 2, 'conv3': 1, 'conv5': 2, 'conv6': 1, 'conv8': 3, 'conv9': 2, 'conv13': 3, 'conv14': 2, 'conv18': 2}.get(name, 1)
    tot['fwd'] += t_f * mult; tot['dgrad'] += t_d * mult; tot['wgrad'] += t_w * mult
    f = lambda t: '%7.1f|%4.0f' % (t, fl / t / 1e6) if t > 0 else '      -     '
    print('%-8s %s %s %s' % (name, f(t_f), f(t_d), f(t_w)))
print('network-weighted totals (us): fwd %.0f dgrad %.0f wgrad %.0f  sum %.0f   %s' % (tot['fwd'], tot['dgrad'], tot['wgrad'], sum(tot.values()), tag))

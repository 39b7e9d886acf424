"""Filter gradient of the 3x3 layers, per layer: the per-tap kernel (conv_wgrad.hip, round 4's path) against the row-of-taps kernel
(conv_wgrad3.hip) by rule and with each variant forced; every result is compared with the per-tap kernel's on the same operands.
usage: [B=16] [LAYERS=conv8,conv13] [VARIANTS=2,0,10,11,12] python scripts/wgrad_ab.py [tag]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_tf_amd import ops

LAYERS = [('conv1', 208, 32, 64, 1), ('conv2', 104, 64, 128, 2), ('conv5', 52, 128, 256, 2), ('conv8', 26, 256, 512, 3),
          ('conv13', 13, 512, 1024, 3), ('conv18', 13, 1024, 1024, 2), ('conv20', 13, 3072, 1024, 1)]
B = int(os.environ.get('B', 16))
SIZE = int(os.environ.get('SIZE', 416))      # network input size: the layers' image sizes scale with it (multi-scale training: 320 .. 608)
LAYERS = [(l[0], l[1] * SIZE // 416) + tuple(l[2:]) for l in LAYERS]
if os.environ.get('LAYERS'):
    LAYERS = [l for l in LAYERS if l[0] in os.environ['LAYERS'].split(',')]
VARIANTS = [int(v) for v in os.environ.get('VARIANTS', '2,0,10,11,12').split(',')]
NAMES = {2: 'per-tap', 0: 'rule', 10: 'row v0', 11: 'row v1', 12: 'row v2', 13: 'row v3', 14: 'row v4', 15: 'row v5'}
T = torch.bfloat16
tag = sys.argv[1] if len(sys.argv) > 1 else ''


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
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 60 * 1e3


print('filter gradient, batch %d, us | TFLOP/s | plan BC/BN/waves/taps/ranges/remap/blocks/direct | max |diff| vs per-tap / max |ref|   %s' % (B, tag))
tot = {v: 0.0 for v in VARIANTS}
for name, H, cin, cout, mult in LAYERS:
    M = B * H * H
    gen = torch.Generator(device='cuda').manual_seed(H + cin)
    x = torch.randn(M * cin, device='cuda', generator=gen).to(T)
    dy = torch.randn(M * cout, device='cuda', generator=gen).to(T)
    flops = 2.0 * M * cin * cout * 9
    ref = None
    row = '%-7s' % name
    for v in VARIANTS:
        ops.set_wgrad_variant(v)
        try:
            dW = torch.zeros(9 * cin * cout, dtype=torch.float32, device='cuda')
            ops.conv2d_wgrad(x, dy, dW, B, H, H, cin, cin, cout, cout, 3)
            plan = ops.last_wgrad_plan()
            torch.cuda.synchronize()
            if v == 2:
                ref = dW.clone()
            err = float((dW - ref).abs().max() / ref.abs().max()) if ref is not None else float('nan')
            t = timeit(lambda: ops.conv2d_wgrad(x, dy, dW, B, H, H, cin, cin, cout, cout, 3))
            tot[v] += t * mult
            row += '  %-7s %6.1f|%5.0f %s %.1e' % (NAMES.get(v, str(v)), t, flops / t / 1e6, '/'.join(str(p) for p in plan.values()), err)
        except Exception as e:      # noqa: BLE001
            row += '  %-7s failed: %s' % (NAMES.get(v, str(v)), str(e)[:60])
        finally:
            ops.set_wgrad_variant(0)
    print(row, flush=True)
print('network-weighted totals (us): ' + '  '.join('%s %.0f' % (NAMES.get(v, str(v)), t) for v, t in tot.items()), tag)

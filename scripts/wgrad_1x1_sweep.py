"""Filter gradient of the 1x1 layers (per-tap kernel, 64 x 64 tiles, pixel ranges summed with f32 atomics) against the block target of its plan:
YOLO2_WGRAD_BLOCKS is read once per process -- run one process per value.  Cold = buffer sets rotated beyond L2.  usage: YOLO2_WGRAD_BLOCKS=256 python scripts/wgrad_1x1_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_tf_amd import ops
T = torch.bfloat16
B = int(os.environ.get('B', 16))
LAYERS = [('conv3', 104, 128, 64), ('conv6', 52, 256, 128), ('conv9_11', 26, 512, 256), ('conv14_16', 13, 1024, 512), ('conv21', 26, 512, 64), ('convout', 13, 1024, 125)]


def timed(fns, n=24):
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for i in range(n):
                fns[i % len(fns)]()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n * 1e3)
    return best


out = []
for name, H, cin, cout in LAYERS:
    M = B * H * H
    ldx, ldy = ops.pad8(cin), ops.pad8(cout)
    nset = 4
    xs = [torch.randn(M * ldx, device='cuda').to(T) for _ in range(nset)]
    dys = [torch.randn(M * ldy, device='cuda').to(T) for _ in range(nset)]
    dW = torch.zeros(cin * cout, dtype=torch.float32, device='cuda')
    mk = lambda i: (lambda: ops.conv2d_wgrad(xs[i], dys[i], dW, B, H, H, cin, ldx, cout, ldy, 1))
    t = timed([mk(i) for i in range(nset)])
    p = ops.last_wgrad_plan()
    out.append('%s %.1f us (%d ranges, %d blocks)' % (name, t, p['ranges'], p['blocks']))
print('YOLO2_WGRAD_BLOCKS=%s batch %d: %s' % (os.environ.get('YOLO2_WGRAD_BLOCKS', 'rule'), B, ';  '.join(out)), flush=True)

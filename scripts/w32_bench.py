"""conv1's filter gradient (32 -> 64 channels at 208 x 208): conv_wgrad_c32.hip (all nine taps per workgroup, every byte once) against the per-tap kernel's
tap-pair form (forced with the wgrad variant switch), warm (one buffer set, hipGraph replay) and cold (buffer sets rotated beyond L2 + MALL), at
batch 8 / 16 / 32 and at the multi-scale widths.  python scripts/w32_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_tf_amd import ops
T = torch.bfloat16


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


cases = [tuple(int(v) for v in c.split('x')) for c in os.environ['CASES'].split(',')] if 'CASES' in os.environ else [(16, 208), (8, 208), (32, 208), (8, 160), (8, 240), (8, 304)]
for B, H in cases:
    M = B * H * H
    nset = max(1, min(6, int(640e6 // (M * 192)) + 1))
    xs = [torch.randn(M * 32, device='cuda').to(T) for _ in range(nset)]
    dys = [torch.randn(M * 64, device='cuda').to(T) for _ in range(nset)]
    dW = torch.zeros(9 * 32 * 64, dtype=torch.float32, device='cuda')
    out = []
    for name, var in ((('nine-tap', 0),) if 'YOLO2_W32_ABL' in os.environ else (('nine-tap', 0), ('per-tap', 2))):
        ops.set_wgrad_variant(var)
        try:
            mk = lambda i: (lambda: ops.conv2d_wgrad(xs[i], dys[i], dW, B, H, H, 32, 32, 64, 64, 3))
            warm = timed([mk(0)])
            plan = ops.last_wgrad_plan()
            cold = timed([mk(i) for i in range(nset)])
        finally:
            ops.set_wgrad_variant(0)
        out.append('%s warm %.1f / cold %.1f us (%.2f TB/s of 192 B per pixel cold; plan %s)' % (
            name, warm, cold, M * 192 / cold * 1e-6, '/'.join(str(plan[k]) for k in ('BC', 'BN', 'waves', 'pair', 'ranges', 'blocks'))))
    print('%sbatch %d, %d x %d (%d buffer sets): %s' % ('ABL %s ' % os.environ['YOLO2_W32_ABL'] if 'YOLO2_W32_ABL' in os.environ else '', B, H, H, nset, ';  '.join(out)), flush=True)

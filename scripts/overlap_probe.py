"""Can a bandwidth-bound kernel without LDS run UNDER the ping-pong convolution?  The plain Adam kernel (41 VGPRs, no LDS: 28 bytes per parameter) on a
second stream against conv20's forward / conv18's data gradient (256 workgroups, the whole LDS of every CU, 216 VGPRs x 2 waves per SIMD) on the first:
wall time of both together against the sum of each alone.  usage: python scripts/overlap_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_tf_amd import ops
T = torch.bfloat16
B = 16
ws = torch.zeros(1024 + 256 * 256 * 128, dtype=torch.float32, device='cuda')
NP = int(os.environ.get('PARAMS', 16 * 1024 * 1024))
w = torch.zeros(NP, device='cuda'); g = torch.ones(NP, device='cuda') * 1e-3; m = torch.zeros(NP, device='cuda'); v = torch.zeros(NP, device='cuda')
side = torch.cuda.Stream(priority=int(os.environ.get('PRIO', '0')))


def wall(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, H, cin, cout in (('conv20 forward', 13, 3072, 1024), ('conv18 forward', 13, 1024, 1024), ('conv8 forward', 26, 256, 512)):
    M = B * H * H
    x = torch.randn(M * cin, device='cuda').to(T)
    y = torch.zeros(M * cout, dtype=T, device='cuda')
    F = torch.zeros(cout * 9 * cin, dtype=T, device='cuda')
    ops.filter_prep(torch.randn(9 * cin * cout, device='cuda') * 0.05, F, None, 3, cin, cin, cout, cout, T)
    NC = 8

    def convs():
        for _ in range(NC):
            ops.conv2d_ws(x, F, None, y, ws, B, H, H, cin, cin, cout, cout, 3)

    def adam():
        ops.adam(w, g, m, v, NP, 1e-6, 0.9, 0.999, 1e-8)

    def both():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            adam()
        convs()
        cur.wait_stream(side)

    t_c, t_a, t_b = wall(convs), wall(adam), wall(both)
    print('%-15s x %d: %.0f us alone;  Adam of %d M parameters (%.2f GB): %.0f us alone = %.2f TB/s;  together %.0f us  (sum %.0f, hidden %.0f %% of the shorter)' % (
        name, NC, t_c, NP >> 20, NP * 28e-9, t_a, NP * 28 / t_a * 1e-6, t_b, t_c + t_a, 100 * (t_c + t_a - t_b) / min(t_c, t_a)))

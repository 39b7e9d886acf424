"""Where a launch of the ping-pong 3x3 kernel spends its time OUTSIDE the K loop (conv_pp.hip SCHED +8192, experiments build): wall-clock
(100 MHz) sums per phase of waves 0 and 4 of every workgroup -- prologue, K loop, park (tail stores issued), flag waits, partner reads, drain,
staging, store issue, store acknowledgement, final publication -- plus the start / end skew of the grid.
usage: YOLO2_LIB_PATH=$PWD/yolo_tf_amd/csrc/libyolo2hip_exp.so [B=16] [LAYERS=conv8,conv13] [DGRAD=1] python scripts/pp_fixed_cost.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from yolo_tf_amd import ops, _lib

LAYERS = [('conv5', 52, 128, 256), ('conv8', 26, 256, 512), ('conv13', 13, 512, 1024), ('conv18', 13, 1024, 1024), ('conv20', 13, 3072, 1024)]
if os.environ.get('LAYERS'):
    LAYERS = [l for l in LAYERS if l[0] in os.environ['LAYERS'].split(',')]
B = int(os.environ.get('B', 16))
DGRAD = os.environ.get('DGRAD', '0') == '1'
T = torch.bfloat16
ws = torch.zeros(1024 + 256 * 256 * 128, dtype=torch.float32, device='cuda')
NAMES = ['prologue', 'K loop', 'park', 'flag wait', 'partner read', 'drain', 'staging', 'store issue', 'final publish', 'segments', 't_begin', 't_end', 'store ack']
lib = _lib.load()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print('batch %d, %s; microseconds (100 MHz wall clock), mean over workgroups [max], wave 0 | wave 4 (S4=1: computing wave 0 | loader wave 4)' % (B, 'data gradient' if DGRAD else 'forward + statistics'))
for name, H, cin, cout in LAYERS:
    if DGRAD:
        cin, cout = cout, cin
    M = B * H * H
    x = torch.randn(M * cin, device='cuda').to(T)
    y = torch.zeros(M * cout, dtype=T, device='cuda')
    w = torch.randn(9 * cin * cout, device='cuda') * 0.05
    Ff = torch.zeros(cout * 9 * cin, dtype=T, device='cuda')
    ops.filter_prep(w, Ff, None, 3, cin, cin, cout, cout, T)
    part = torch.zeros(2 * 256 * cout, dtype=torch.float32, device='cuda')
    shift = torch.zeros(cout, device='cuda')
    BN = os.environ.get('BN', '0') == '1'      # DGRAD=1 BN=1: the data gradient with the producer's BN-backward sums in its epilogue
    yprev = torch.randn(M * cout, device='cuda').to(T)
    mean_, var_ = torch.zeros(cout, device='cuda'), torch.ones(cout, device='cuda')
    dg, db = torch.zeros(cout, device='cuda'), torch.zeros(cout, device='cuda')
    red = torch.zeros(ops.workspace_bytes('bn', cout) // 8, dtype=torch.float64, device='cuda')
    fn = (lambda: ops.conv2d_dgrad_bn(x, Ff, y, ws, B, H, H, cin, cin, cout, cout, 3, yprev, mean_, var_, mean_ + 1, mean_, dg, db, part, red, 1e-5, 0.1)) if (DGRAD and BN) else \
         (lambda: ops.conv2d_ws(x, Ff, None, y, ws, B, H, H, cin, cin, cout, cout, 3)) if DGRAD else \
         (lambda: ops.conv2d_bn(x, Ff, y, ws, B, H, H, cin, cin, cout, cout, 3, shift, part))
    S4 = os.environ.get('S4', '0') == '1'      # the loader / consumer kernel (conv_s4.hip; experiments build with s4)
    ops.set_igemm_tap(3 if S4 else 2)
    ops.set_pp_cost(int(os.environ.get('CV', 0)))
    ops.set_pp(grid=0, dmapos=2, min_steps=-1, min_share=-1)
    t_prod = timed(fn)
    plan = ops.last_conv_plan()
    if S4:
        lib.yolo2_debug_set_s4_abl(32 + int(os.environ.get('ABL', 0)))
    else:
        ops.set_pp(grid=0, dmapos=2 + 8192, min_steps=-1, min_share=-1)
    t_st = timed(fn)
    if S4:
        lib.yolo2_debug_set_s4_abl(0)
    torch.cuda.synchronize()
    G = plan['grid_x']
    buf = np.zeros(1024 * 2 * 16, np.uint64)
    rc = (lib.yolo2_debug_s4_phases if S4 else lib.yolo2_debug_pp_phases)(ctypes.c_void_p(buf.ctypes.data))
    assert rc == 0, rc
    d = buf.reshape(1024, 2, 16)[:G].astype(np.float64) / 100.0      # us
    t0, t1 = d[:, :, 10].min(), d[:, :, 11].max()
    print('%-7s grid %3d  product %.1f us, stamped %.1f us; device span first begin -> last end %.1f us; begin skew %.2f us, end skew %.2f us; segments per workgroup %.2f' % (
        name, G, t_prod, t_st, t1 - t0, d[:, 0, 10].max() - t0, t1 - d[:, 0, 11].min(), d[:, 0, 9].mean() * 100))
    for k in (0, 1, 2, 3, 4, 5, 6, 7, 12, 8):
        a, b_ = d[:, 0, k], d[:, 1, k]
        print('        %-14s %6.2f [%6.2f] | %6.2f [%6.2f]' % (NAMES[k], a.mean(), a.max(), b_.mean(), b_.max()))
    own = d[:, 0, 3] + d[:, 0, 4] > 0
    if own.any():
        print('        owners (%d workgroups): flag wait %.2f [%.2f], partner read %.2f [%.2f]; lifetime of a workgroup %.2f [%.2f]' % (
            own.sum(), d[own, 0, 3].mean(), d[own, 0, 3].max(), d[own, 0, 4].mean(), d[own, 0, 4].max(), (d[:, 0, 11] - d[:, 0, 10]).mean(), (d[:, 0, 11] - d[:, 0, 10]).max()))
    if os.environ.get('DETAIL'):
        # per workgroup (wave 0): begin, the phases in order, end -- the latest finishers first, then every 16th workgroup
        end = d[:, 0, 11] - t0
        order = list(np.argsort(-end)[:int(os.environ['DETAIL'])]) + list(range(0, G, 16))
        print('        wx   begin  prolog   loop   park  fwait  pread  drain  stage  store   ack   fpub  segs    end')
        for wxi in order:
            r = d[wxi, 0]
            print('        %3d %6.2f %7.2f %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f %5.2f %6.2f %5.0f %6.2f' % (
                wxi, r[10] - t0, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[12], r[8], r[9] * 100, r[11] - t0))
ops.set_pp(grid=0, dmapos=2, min_steps=18, min_share=24)
ops.set_igemm_tap(2)

"""Parity at the BENCHMARKED launch shapes, through the C ABI, against the CPU oracle.

Kernel-variant selection is shape-driven everywhere (conv_igemm.hip launch_conv, conv_wgrad.hip wgrad_plan), so
toy-shape parity does not transfer: this file runs every distinct convolution of Darknet-19 YOLOv2 @416 (reference
model/yolo2/inference.py:70-118) at the two shapes the benchmarks use -- batch 16 VOC-20 (BASELINE configs[1]) and batch 8
COCO-80, the per-GPU shape of configs[2] (425-wide head) -- forward (with the batch-norm statistics epilogue, as the engine
calls it), data gradient and filter gradient, in bf16 (the benchmarked dtype) with the engine's real 33.5 MB workspace, and
compares with oracle.conv2d / conv2d_dgrad / conv2d_wgrad on the same bf16-rounded operands.  Each test also records which
variant ran (yolo2_debug_last_*_plan) and asserts it for the launches that carry the benchmark: the 256x128 stream-K tile
(conv18/19/20 forward + data gradient) and the XCD-local atomic filter-gradient plan (26x26 stages).

Tolerance: operands are exact in bf16, products exact in f32, accumulation f32 on both sides -> the only differences are the
summation order (~1e-6 of the output scale) and, for activations, the final bf16 rounding of the stored output
(<= 2^-9 relative per element): |got - ref| <= 4e-3 |ref| + 2e-4 max|ref| per element; filter gradients (f32 out,
f32 atomics over up to 173k pixels) against the f64 sum of the same exact products, per element: 2e-5 |ref| + 4e-6 max|ref|
(the f32 summation-order noise measures 0.4-1.3e-6 of the scale; a dropped border column of a padded-index kernel leaves 1e-2)."""
import os

import numpy as np
import pytest
import torch

TAP_STAGES = 3 if os.environ.get('YOLO2_IGEMM_TAP', '2') == '0' else 18      # plan word 'stages' of the kernel that takes the 13x13 layers: ping-pong tap-fused (default) / per-tap stream-K (A/B)

from oracle import yolo2_ref as R

pytestmark = pytest.mark.gpu

# name, H (= W), Cin, Cout, ksize, bn  -- one entry per distinct launch shape of the 22 convolutions
LAYERS = [
    ('conv0', 416, 3, 32, 3, True), ('conv1', 208, 32, 64, 3, True), ('conv2_4', 104, 64, 128, 3, True),
    ('conv3', 104, 128, 64, 1, True), ('conv5_7', 52, 128, 256, 3, True), ('conv6', 52, 256, 128, 1, True),
    ('conv8_10_12', 26, 256, 512, 3, True), ('conv9_11', 26, 512, 256, 1, True), ('conv13_15_17', 13, 512, 1024, 3, True),
    ('conv14_16', 13, 1024, 512, 1, True), ('conv18_19', 13, 1024, 1024, 3, True), ('conv20', 13, 3072, 1024, 3, True),
    ('conv_out', 13, 1024, None, 1, False),
]
CONFIGS = [('b16_voc20', 16, 125), ('b8_coco80', 8, 425)]
WS_FLOATS = 1024 + 256 * 256 * 128          # engine.Engine.conv_ws


@pytest.fixture(scope='module')
def ops():
    from yolo_tf_amd import ops as _ops
    _ops._lib.load()
    assert torch.cuda.is_available()
    return _ops


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy()


def dev_bf16(a, ld):
    out = np.zeros(a.shape[:-1] + (ld,), np.float32)
    out[..., :a.shape[-1]] = a
    return torch.from_numpy(out).to('cuda').to(torch.bfloat16).contiguous()


def host(t):
    return t.float().cpu().numpy()


def check_act(got, ref, what):
    scale = float(np.abs(ref).max())
    bad = np.abs(got - ref) > 4e-3 * np.abs(ref) + 2e-4 * scale
    assert not bad.any(), '%s: %d of %d elements off, worst abs err %.3e at scale %.3e' % (
        what, int(bad.sum()), bad.size, float(np.abs(got - ref).max()), scale)


def _cases():
    for cname, B, head in CONFIGS:
        for name, H, cin, cout, k, bn in LAYERS:
            yield pytest.param(cname, B, name, H, cin, cout if cout else head, k, bn, id='%s-%s' % (cname, name))


def _inputs(B, H, cin, cout, k, seed):
    rng = np.random.RandomState(seed)
    x = bf16_round(rng.randn(B, H, H, cin).astype(np.float32))
    w = bf16_round((rng.randn(k, k, cin, cout) / np.sqrt(k * k * cin)).astype(np.float32))
    dy = bf16_round(rng.randn(B, H, H, cout).astype(np.float32))
    return x, w, dy


@pytest.mark.parametrize('cname,B,name,H,cin,cout,k,bn', list(_cases()))
def test_forward_bench_shape(ops, cname, B, name, H, cin, cout, k, bn):
    x, w, _ = _inputs(B, H, cin, cout, k, 1000 + H + cin + cout + B)
    ldx, ldy = ops.pad8(cin), ops.pad8(cout)
    xd = dev_bf16(x, ldx)
    F = torch.zeros(cout * k * k * ldx, dtype=torch.bfloat16, device='cuda')
    ops.filter_prep(torch.from_numpy(w).cuda(), F, None, k, cin, ldx, cout, ldy, torch.bfloat16)
    ws = torch.full((WS_FLOATS,), 7.0, dtype=torch.float32, device='cuda')            # dirty on purpose
    M = B * H * H
    O = torch.zeros(M * ldy, dtype=torch.bfloat16, device='cuda')
    if bn:
        # the engine's training forward: statistics from the convolution's own epilogue (engine.Engine._conv)
        assert ldy == cout
        part = torch.zeros(2 * 256 * cout, dtype=torch.float32, device='cuda')
        shift = torch.zeros(cout, dtype=torch.float32, device='cuda')
        mean, var = torch.zeros(cout, device='cuda'), torch.zeros(cout, device='cuda')
        ops.conv2d_bn(xd, F, O, ws, B, H, H, ldx, ldx, cout, ldy, k, shift, part)
        plan = ops.last_conv_plan()
        ops.bn_finalize(part, shift, M, cout, mean, var, None, None, 0.999)
    else:
        bias = torch.from_numpy(np.random.RandomState(3).randn(cout).astype(np.float32)).cuda()
        ops.conv2d_ws(xd, F, bias, O, ws, B, H, H, ldx, ldx, cout, ldy, k)
        plan = ops.last_conv_plan()
    torch.cuda.synchronize()
    print('\nPLAN fwd %s %s: %s' % (cname, name, plan))
    y = host(O).reshape(B, H, H, ldy)
    assert np.all(y[..., cout:] == 0), 'padding lanes must stay untouched'
    ref = R.conv2d(x, w)
    if not bn:
        ref = ref + host(bias)
    check_act(y[..., :cout], ref, 'forward %s %s' % (cname, name))
    if bn:
        y64 = y[..., :cout].reshape(M, cout).astype(np.float64)
        m_ref, v_ref = y64.mean(0), y64.var(0)
        assert np.abs(host(mean) - m_ref).max() <= 2e-5 * np.sqrt(v_ref).max() + 1e-6
        assert np.abs(host(var) - v_ref).max() <= 1e-4 * v_ref.max()
        assert float(part.abs().max()) == 0.0
    if name == 'conv2_4' and 'YOLO2_C64' not in os.environ:
        # 64 -> 128 channels at 104 x 104: the persistent filter-in-registers kernel (conv_c64.hip), statistics spread over its 32 partial rows
        assert (plan['BM'], plan['BN'], plan['stages']) == (256, 128, 9) and plan['grid_x'] == 32, plan
    if name == 'conv1' and 'YOLO2_C32' not in os.environ:
        assert (plan['BM'], plan['BN'], plan['stages']) == (512, 64, 9), plan
    if name in ('conv18_19', 'conv20') and B == 16:
        # the launches that carry the benchmark: tap-fused 256x128 stream-K, one workgroup per CU ('stages' 9 = nine taps per halo image)
        assert plan['BM'] == 256 and plan['split'] == 2 and plan['waves'] == 8 and plan['stages'] == TAP_STAGES, plan
    if name in ('conv18_19', 'conv20') and B == 8 and TAP_STAGES == 18 and 'YOLO2_PP_LONG_SHARE' not in os.environ:
        # batch 8: 48 tiles cut into 5.3 shares of 27 / 81 K steps -- the long-share clause of the launch rule (profiles/r05_long_share_b8.txt)
        assert (plan['BM'], plan['stages'], plan['split']) == (256, 18, 2), plan
    if name == 'conv13_15_17' and B == 16:
        # 88 tiles of 256 x 128, 24.75 K steps per workgroup, a tile cut into ~3 shares: stream-K on the ping-pong kernel (per-tap 128x128 stream-K with YOLO2_IGEMM_TAP=0)
        assert plan['split'] == 2 and (plan['BM'], plan['stages']) == ((256, 18) if TAP_STAGES == 18 else (128, 3)), plan
    if name == 'conv8_10_12' and B == 16 and TAP_STAGES == 18:
        assert (plan['BM'], plan['stages'], plan['grid_x']) == (256, 18, 172), plan      # ping-pong kernel, one workgroup per tile (67 % of the CUs), no hand-off
    if name == 'conv1' and 'YOLO2_C32' not in os.environ:
        assert (plan['BM'], plan['BN'], plan['stages']) == (512, 64, 9), plan      # 32 -> 64 channels: the persistent kernel of conv_c32.hip (512-position tiles)


@pytest.mark.parametrize('cname,B,name,H,cin,cout,k,bn', [c for c in _cases() if c.values[2] != 'conv0'])
def test_dgrad_bench_shape(ops, cname, B, name, H, cin, cout, k, bn):
    _, w, dy = _inputs(B, H, cin, cout, k, 2000 + H + cin + cout + B)
    ldx, ldy = ops.pad8(cin), ops.pad8(cout)
    dyd = dev_bf16(dy, ldy)
    F = torch.zeros(cin * k * k * ldy, dtype=torch.bfloat16, device='cuda')
    ops.filter_prep(torch.from_numpy(w).cuda(), None, F, k, cin, ldx, cout, ldy, torch.bfloat16)
    ws = torch.full((WS_FLOATS,), -3.0, dtype=torch.float32, device='cuda')
    dx = torch.zeros(B * H * H * ldx, dtype=torch.bfloat16, device='cuda')
    ops.conv2d_ws(dyd, F, None, dx, ws, B, H, H, ldy, ldy, cin, ldx, k)
    plan = ops.last_conv_plan()
    torch.cuda.synchronize()
    print('\nPLAN dgrad %s %s: %s' % (cname, name, plan))
    got = host(dx).reshape(B, H, H, ldx)
    assert np.all(got[..., cin:] == 0)
    check_act(got[..., :cin], R.conv2d_dgrad(dy, w), 'dgrad %s %s' % (cname, name))
    if name in ('conv18_19', 'conv20') and B == 16:
        assert plan['BM'] == 256 and plan['split'] == 2 and plan['stages'] == TAP_STAGES, plan
    if name == 'conv5_7' and B == 16 and TAP_STAGES == 18:
        assert (plan['BM'], plan['stages'], plan['grid_x']) == (256, 18, 169), plan      # 52x52 data gradient: 169 whole tiles
    if name == 'conv1' and 'YOLO2_C64' not in os.environ:
        assert (plan['BM'], plan['BN'], plan['stages']) == (256, 32, 9), plan      # 64 -> 32 at 208 x 208: the 32-filter form of conv_c64.hip
    if name == 'conv2_4' and B == 16 and 'YOLO2_C64' not in os.environ:
        assert (plan['BM'], plan['BN'], plan['stages']) == (128, 64, 9), plan      # 128 -> 64 at 104 x 104: the two-channel-half form of conv_c64.hip


@pytest.mark.parametrize('cname,B,name,H,cin,cout,k,bn', list(_cases()))
def test_wgrad_bench_shape(ops, cname, B, name, H, cin, cout, k, bn):
    x, _, dy = _inputs(B, H, cin, cout, k, 3000 + H + cin + cout + B)
    ldx, ldy = ops.pad8(cin), ops.pad8(cout)
    accumulates = ops.conv2d_wgrad_accumulates(B, H, H, cin, ldx, cout, ldy, k, torch.bfloat16)
    # single-range plans overwrite: hand them a dirty buffer, like the engine (which only zeroes accumulating layers)
    dW = torch.zeros(k * k * cin * cout, dtype=torch.float32, device='cuda') if accumulates else \
        torch.full((k * k * cin * cout,), 99.0, dtype=torch.float32, device='cuda')
    ops.conv2d_wgrad(dev_bf16(x, ldx), dev_bf16(dy, ldy), dW, B, H, H, cin, ldx, cout, ldy, k)
    plan = ops.last_wgrad_plan()
    torch.cuda.synchronize()
    print('\nPLAN wgrad %s %s: %s accumulates=%s' % (cname, name, plan, accumulates))
    ref = R.conv2d_wgrad(x.astype(np.float64), dy.astype(np.float64), k, k)      # exact operands and products: the f64 sum is THE answer
    got = host(dW).reshape(k, k, cin, cout).astype(np.float64)
    scale = float(np.abs(ref).max())
    ratio = np.abs(got - ref) / (2e-5 * np.abs(ref) + 4e-6 * scale)
    print('wgrad %s %s: worst |err| / (2e-5 |ref| + 4e-6 scale) = %.3f; max |err| = %.2e of the scale' % (cname, name, ratio.max(), np.abs(got - ref).max() / scale))
    assert ratio.max() <= 1.0, 'wgrad %s %s: %d elements beyond 2e-5 rel + 4e-6 of scale %.3e (worst ratio %.2f)' % (cname, name, int((ratio > 1).sum()), scale, ratio.max())
    if plan['BC'] > 0:
        assert bool(plan['direct']) == (not accumulates), plan
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    if 'YOLO2_WGRAD_VARIANT' not in os.environ and 'YOLO2_W3_KS' not in os.environ:
        # the launches that carry the benchmark's filter gradients: the row-of-taps kernel (conv_wgrad3.hip; plan word 'pair' == 3: three taps per workgroup)
        if name in ('conv13_15_17', 'conv20'):
            assert (plan['pair'], plan['BC'], plan['BN'], plan['direct'], plan['ranges']) == (3, 64, 128, 1, 1), plan      # two wave groups, plain stores
        if name == 'conv18_19':
            assert (plan['pair'], plan['BC'], plan['BN'], plan['direct']) == (3, 128, 128, 1), plan                       # 192 workgroups of 128 x 128
        if name in ('conv2_4', 'conv5_7', 'conv8_10_12'):
            assert (plan['pair'], plan['BC'], plan['BN'], plan['direct']) == (3, 64, 64, 0), plan                         # split reduction, summed per workgroup
            assert plan['blocks'] <= 3 * cus // 2 and plan['ranges'] * 3 * (cin // 64) * (cout // 64) <= cus, plan           # at most one workgroup per CU
        if name == 'conv1' and 'YOLO2_WGRAD_C32' not in os.environ:
            assert (plan['pair'], plan['BC'], plan['BN'], plan['waves'], plan['direct']) == (9, 32, 64, 12, 0), plan      # 32 input channels: all nine taps per workgroup (conv_wgrad_c32.hip)
            assert plan['blocks'] <= cus, plan


@pytest.mark.parametrize('name,cin', [('conv13_15_17', 512), ('conv18_19', 1024), ('conv20', 3072)])
def test_streamk_under_concurrent_row_wgrad(ops, name, cin):
    """The product pairing of the backward pass (engine.Engine.backward): the stream-K forward / data-gradient launches hand partial tiles
    between workgroups through flags on the main stream while the row-of-taps filter gradient (conv_wgrad3.hip: one ~150 KB-LDS workgroup per
    CU) runs on a side stream and takes CUs the stream-K launch counted on.  A residency problem now gives a bounded wait and WRONG DATA, not a
    hang, so: repeat the pair a few times with both in flight, check every element of all three results every time, and ask the device for
    give-ups after each iteration."""
    B, H, cout, k = 16, 13, 1024, 3
    x, w, dy = _inputs(B, H, cin, cout, k, 4000 + cin)
    ldx, ldy = cin, cout
    xd, dyd = dev_bf16(x, ldx), dev_bf16(dy, ldy)
    Ff = torch.zeros(cout * k * k * ldx, dtype=torch.bfloat16, device='cuda')
    Fd = torch.zeros(cin * k * k * ldy, dtype=torch.bfloat16, device='cuda')
    ops.filter_prep(torch.from_numpy(w).cuda(), Ff, Fd, k, cin, ldx, cout, ldy, torch.bfloat16)
    ws = torch.zeros(WS_FLOATS, dtype=torch.float32, device='cuda')
    ref_y, ref_dx = R.conv2d(x, w), R.conv2d_dgrad(dy, w)
    ref_dw = R.conv2d_wgrad(x.astype(np.float64), dy.astype(np.float64), k, k)
    scale_dw = float(np.abs(ref_dw).max())
    side = torch.cuda.Stream()
    for it in range(4):
        O = torch.zeros(B * H * H * ldy, dtype=torch.bfloat16, device='cuda')
        dx = torch.zeros(B * H * H * ldx, dtype=torch.bfloat16, device='cuda')
        dW = torch.full((k * k * cin * cout,), float(it), dtype=torch.float32, device='cuda')      # (13x13: single range, plain stores over a dirty buffer)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            ops.conv2d_wgrad(xd, dyd, dW, B, H, H, cin, ldx, cout, ldy, k)
            assert ops.last_wgrad_plan()['pair'] == 3, ops.last_wgrad_plan()
        ops.conv2d_ws(dyd, Fd, None, dx, ws, B, H, H, ldy, ldy, cin, ldx, k)
        plan_d = ops.last_conv_plan()
        ops.conv2d_ws(xd, Ff, None, O, ws, B, H, H, ldx, ldx, cout, ldy, k)
        plan_f = ops.last_conv_plan()
        assert plan_f['split'] == 2 and plan_f['grid_x'] > 88, plan_f            # stream-K: more workgroups than tiles
        if cin >= 1024:
            assert plan_d['BM'] == 256 and plan_d['split'] == 2, plan_d
        side.synchronize()
        ops.check_async_errors()                                                  # (synchronises the main stream; raises on a give-up)
        check_act(host(O).reshape(B, H, H, cout), ref_y, 'forward beside the row filter gradient, iteration %d' % it)
        check_act(host(dx).reshape(B, H, H, cin), ref_dx, 'dgrad beside the row filter gradient, iteration %d' % it)
        got = host(dW).reshape(k, k, cin, cout).astype(np.float64)
        assert (np.abs(got - ref_dw) <= 2e-5 * np.abs(ref_dw) + 4e-6 * scale_dw).all(), 'filter gradient beside stream-K, iteration %d' % it

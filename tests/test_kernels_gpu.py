"""Per-kernel parity: every C-ABI entry point vs the CPU oracle on the same seeded inputs.
Bit-exact for moves / integer decisions (reorg, pool routing, NMS keep-sets, filter layouts);
1e-4 relative for fp32 arithmetic (BASELINE.json north_star); bf16 mode is compared with the
oracle run on the bf16-rounded inputs at a bf16-sized tolerance (stated per test)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import yolo2_ref as R

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32_RTOL = 1e-4          # north_star: within 1e-4 rel for fp32 conv/loss
BF16_RTOL = 2e-2         # bf16 storage (8 mantissa bits) of outputs + bf16 operands, relative to the output scale


@pytest.fixture(scope='module')
def ops():
    from yolo_tf_amd import ops as _ops
    _ops._lib.load()
    assert torch.cuda.is_available()
    return _ops


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to('cuda').to(dtype).contiguous()


def host(t):
    return t.float().cpu().numpy()


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy()


WORST = {}      # what -> worst measured error of the run, in units of the bound (printed per test; shown with pytest -s or on failure)


def assert_close(got, ref, rtol, what='', per_element=None):
    """fp32 claims against the oracle (rtol == F32_RTOL, north_star "within 1e-4 rel"): PER ELEMENT |err| <= rtol * |ref| + 0.1 * rtol * max|ref| -- relative where
    the reference is not small, with a floor of a tenth of the tolerance at the output scale for elements that are sums cancelling to ~0 (the
    f32 summation order of a 27k-term reduction moves those by ~1e-6 of the scale).  bf16-sized tolerances stay max-norm: the stored
    output's rounding is relative to each element and is tested per element in tests/test_bench_shapes_gpu.py; tighter tolerances (1e-6:
    GPU path against GPU path, same products) are max-norm bounds on the summation order, and so are whole-network comparisons
    (per_element=False: 22 layers deep, a logit that cancels to ~0 carries the absolute error of its largest summand)."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    scale = np.abs(ref).max() + 1e-30
    if (rtol == F32_RTOL) if per_element is None else per_element:
        bound = rtol * np.abs(ref) + 0.1 * rtol * scale
        ratio = np.abs(got - ref) / bound
        worst = float(ratio.max()) if ratio.size else 0.0
        WORST[what] = worst
        print('%s: worst |err| / (%.0e |ref| + %.0e scale) = %.3f' % (what, rtol, 0.1 * rtol, worst))
        assert worst <= 1.0, '%s: %d of %d elements beyond %.0e rel + %.0e of scale %.3e; worst ratio %.2f' % (
            what, int((ratio > 1).sum()), ratio.size, rtol, 0.1 * rtol, scale, worst)
        return
    err = np.abs(got - ref).max() if got.size else 0.0
    print('%s: max abs err %.3e = %.3e of the scale (bound %.1e)' % (what, err, err / scale, rtol))
    assert err <= rtol * scale, '%s: max abs err %.3e vs scale %.3e (rel %.3e > %.1e)' % (what, err, scale, err / scale, rtol)


WGRAD_REL, WGRAD_FLOOR = 2e-5, 4e-6


def assert_wgrad_exact_products(got, ref, what):
    """bf16 filter gradients: exact products, f32 accumulation (MFMA chains, LDS reduction of the wave groups, f32 atomics over the pixel
    ranges), f64 reference.  PER ELEMENT |err| <= 2e-5 |ref| + 4e-6 max|ref|: the floor is the f32 summation-order noise of the longest
    reductions (173k pixels: partial sums ~sqrt(K) carry half an ulp each; measured 0.4-1.3e-6 of the scale), three orders of magnitude below
    what a dropped border column or row of the padded-index kernels would leave (1 / W of an element)."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    scale = np.abs(ref).max() + 1e-30
    ratio = np.abs(got - ref) / (WGRAD_REL * np.abs(ref) + WGRAD_FLOOR * scale)
    worst = float(ratio.max())
    WORST[what] = worst
    print('%s: worst |err| / (%.0e |ref| + %.0e scale) = %.3f; max |err| = %.2e of the scale' % (what, WGRAD_REL, WGRAD_FLOOR, worst, np.abs(got - ref).max() / scale))
    assert worst <= 1.0, '%s: %d of %d elements beyond %.0e rel + %.0e of scale %.3e; worst ratio %.2f' % (
        what, int((ratio > 1).sum()), ratio.size, WGRAD_REL, WGRAD_FLOOR, scale, worst)


def pad_channels(x, ld):
    out = np.zeros(x.shape[:-1] + (ld,), x.dtype)
    out[..., :x.shape[-1]] = x
    return out


def run_conv(ops, x, w, bias, tdtype, ldo=None):
    """x NHWC (np), w HWIO (np): forward through filter_prep + conv2d; returns y NHWC np (Cout real channels)."""
    B, H, W, Cin = x.shape
    k, _, _, Cout = w.shape
    ldp = ops.pad8(Cin)
    ldo = ldo or ops.pad8(Cout)
    xd = dev(pad_channels(x, ldp), tdtype)
    F = torch.zeros(Cout * k * k * ldp, dtype=tdtype, device='cuda')
    ops.filter_prep(dev(w), F, None, k, Cin, ldp, Cout, ops.pad8(Cout), tdtype)
    O = torch.zeros(B * H * W * ldo, dtype=tdtype, device='cuda')
    ops.conv2d(xd, F, None if bias is None else dev(bias), O, B, H, W, ldp, ldp, Cout, ldo, k)
    torch.cuda.synchronize()
    y = host(O).reshape(B, H, W, ldo)
    assert np.all(y[..., Cout:] == 0), 'padding lanes must stay untouched'
    return y[..., :Cout]


CONV_SHAPES = [
    # B, H, W, Cin, Cout, k
    (2, 13, 13, 16, 40, 3),
    (1, 26, 20, 3, 32, 3),      # conv0-like: 3 real channels in an 8-wide pixel stride -> first-layer direct kernels
    (2, 33, 70, 3, 32, 3),      # first layer, ragged 32-pixel segments
    (1, 64, 96, 3, 32, 3),      # first layer, exact segments
    (2, 13, 13, 64, 125, 1),    # final 1x1 + bias, ragged Cout, padded ldo
    (2, 26, 26, 64, 160, 3),    # two N tiles, M tail (1352 rows)
    (1, 8, 8, 24, 64, 3),       # BN=64 tile, channel tail inside a K step
    (3, 5, 7, 136, 72, 3),      # odd extents, several K steps per tap
    (1, 13, 13, 192, 128, 3),   # three 64-channel K chunks forward, two in the data gradient
    (2, 12, 20, 32, 64, 3),     # 32 -> 64 channels: the persistent conv1 kernel (conv_c32.hip), several tiles' worth of padded positions
    (1, 7, 9, 32, 64, 3),       # ... one partial tile, odd extents
    (3, 40, 33, 32, 64, 3),     # ... more tiles than one workgroup round leaves whole
    (2, 9, 11, 320, 200, 3),    # five chunks; data gradient single-chunk (ldy = 200)
    (2, 12, 20, 64, 128, 3),    # 64 -> 128 channels: the persistent conv2 / conv4 kernel with the filter in registers (conv_c64.hip)
    (2, 12, 20, 64, 32, 3),     # 64 -> 32: its 32-filter form (conv1's data gradient)
    (2, 12, 20, 128, 64, 3),    # 128 -> 64: the two-channel-half form (conv2 / conv4's data gradients)
    (1, 7, 9, 64, 128, 3),      # ... one partial tile, odd extents
    (3, 40, 33, 64, 128, 3),    # ... more tiles than one workgroup round leaves whole
]


@pytest.mark.parametrize('shape', CONV_SHAPES)
@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_conv_forward(ops, shape, mode):
    B, H, W, Cin, Cout, k = shape
    rng = np.random.RandomState(sum(shape))
    x = rng.randn(B, H, W, Cin).astype(np.float32)
    w = (rng.randn(k, k, Cin, Cout) / np.sqrt(k * k * Cin)).astype(np.float32)
    bias = rng.randn(Cout).astype(np.float32) if k == 1 else None
    if mode == 'bf16':
        x, w = bf16_round(x), bf16_round(w)
    ref = R.conv2d(x, w) + (0 if bias is None else bias)
    got = run_conv(ops, x, w, bias, torch.float32 if mode == 'f32' else torch.bfloat16)
    assert_close(got, ref, F32_RTOL if mode == 'f32' else BF16_RTOL, 'conv fwd %s %s' % (shape, mode))


@pytest.mark.parametrize('shape', CONV_SHAPES)
@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_conv_dgrad(ops, shape, mode):
    B, H, W, Cin, Cout, k = shape
    tdtype = torch.float32 if mode == 'f32' else torch.bfloat16
    rng = np.random.RandomState(sum(shape) + 1)
    dy = rng.randn(B, H, W, Cout).astype(np.float32)
    w = (rng.randn(k, k, Cin, Cout) / np.sqrt(k * k * Cout)).astype(np.float32)
    if mode == 'bf16':
        dy, w = bf16_round(dy), bf16_round(w)
    ref = R.conv2d_dgrad(dy, w)
    ldy, ldx = ops.pad8(Cout), ops.pad8(Cin)
    dyd = dev(pad_channels(dy, ldy), tdtype)
    F = torch.zeros(Cin * k * k * ldy, dtype=tdtype, device='cuda')
    ops.filter_prep(dev(w), None, F, k, Cin, ldx, Cout, ldy, tdtype)
    dx = torch.zeros(B * H * W * ldx, dtype=tdtype, device='cuda')
    ops.conv2d(dyd, F, None, dx, B, H, W, ldy, ldy, Cin, ldx, k)
    torch.cuda.synchronize()
    got = host(dx).reshape(B, H, W, ldx)[..., :Cin]
    assert_close(got, ref, F32_RTOL if mode == 'f32' else BF16_RTOL, 'conv dgrad %s %s' % (shape, mode))


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(2, 13, 13, 512, 256, 3), (1, 13, 13, 1024, 125, 1), (2, 26, 26, 256, 136, 3),
                                   (8, 13, 13, 1024, 500, 3), (16, 13, 13, 3072, 1000, 1)])   # the last two take the stream-K path
def test_conv_forward_ksliced(ops, shape, mode):
    """yolo2_conv2d_ws: K loop sliced across workgroups (small M x N grids), f32 partial tiles + finishing kernel."""
    B, H, W, Cin, Cout, k = shape
    tdtype = torch.float32 if mode == 'f32' else torch.bfloat16
    rng = np.random.RandomState(sum(shape))
    x = rng.randn(B, H, W, Cin).astype(np.float32)
    w = (rng.randn(k, k, Cin, Cout) / np.sqrt(k * k * Cin)).astype(np.float32)
    bias = rng.randn(Cout).astype(np.float32)
    if mode == 'bf16':
        x, w = bf16_round(x), bf16_round(w)
    ref = R.conv2d(x, w) + bias
    ldo = ops.pad8(Cout)
    F = torch.zeros(Cout * k * k * Cin, dtype=tdtype, device='cuda')
    ops.filter_prep(dev(w), F, None, k, Cin, Cin, Cout, ldo, tdtype)
    O = torch.zeros(B * H * W * ldo, dtype=tdtype, device='cuda')
    ws = torch.full((max(B * H * W * Cout + 64, 1024 + 256 * 128 * 128),), 5.0, dtype=torch.float32, device='cuda')    # dirty on purpose
    ops.conv2d_ws(dev(x, tdtype), F, dev(bias), O, ws, B, H, W, Cin, Cin, Cout, ldo, k)
    torch.cuda.synchronize()
    y = host(O).reshape(B, H, W, ldo)
    assert np.all(y[..., Cout:] == 0)
    assert_close(y[..., :Cout], ref, F32_RTOL if mode == 'f32' else BF16_RTOL, 'conv fwd k-sliced %s %s' % (shape, mode))


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(2, 26, 26, 64, 160, 3),      # direct tiles, M tail
                                   (1, 64, 96, 3, 32, 3),        # first-layer direct kernel (its own statistics epilogue)
                                   (2, 13, 13, 512, 256, 3),     # K-sliced grid -> fallback
                                   (8, 13, 13, 1024, 504, 3),    # stream-K: owners hold the finished tiles
                                   (3, 20, 20, 32, 64, 3),       # 64-filter tile
                                   (2, 13, 13, 128, 256, 1)])
def test_conv_bn_fused_statistics(ops, shape, mode):
    """yolo2_conv2d_bn + yolo2_bn_finalize: same output as yolo2_conv2d_ws, batch moments of the STORED output (tf.nn.moments
    semantics, biased variance), moving averages updated with decay 0.999, partial buffer left zero."""
    B, H, W, Cin, Cout, k = shape
    tdtype = torch.float32 if mode == 'f32' else torch.bfloat16
    rng = np.random.RandomState(sum(shape) + 7)
    x = rng.randn(B, H, W, Cin).astype(np.float32)
    w = (rng.randn(k, k, Cin, Cout) / np.sqrt(k * k * Cin)).astype(np.float32) + 0.02
    ldx = ops.pad8(Cin)
    F = torch.zeros(Cout * k * k * ldx, dtype=tdtype, device='cuda')
    ops.filter_prep(dev(w), F, None, k, Cin, ldx, Cout, Cout, tdtype)
    xd = dev(pad_channels(x, ldx), tdtype)
    ws = torch.full((1024 + 256 * 256 * 128,), 3.0, dtype=torch.float32, device='cuda')
    M = B * H * W
    y_ref = torch.zeros(M * Cout, dtype=tdtype, device='cuda')
    ops.conv2d_ws(xd, F, None, y_ref, ws, B, H, W, ldx, ldx, Cout, Cout, k)
    y = torch.zeros(M * Cout, dtype=tdtype, device='cuda')
    part = torch.zeros(2 * 256 * Cout, dtype=torch.float32, device='cuda')
    mm0 = (rng.randn(Cout) * 0.1).astype(np.float32)
    mv0 = (rng.rand(Cout) + 0.5).astype(np.float32)
    mm, mv = dev(mm0), dev(mv0)
    mean, var = torch.zeros(Cout, device='cuda'), torch.zeros(Cout, device='cuda')
    ops.conv2d_bn(xd, F, y, ws, B, H, W, ldx, ldx, Cout, Cout, k, mm, part)
    ops.bn_finalize(part, mm, M, Cout, mean, var, mm, mv, 0.999)
    torch.cuda.synchronize()
    # (K-sliced grids accumulate with f32 atomics: the summation order, hence the last bit, varies between launches)
    assert_close(host(y), host(y_ref), 1e-6 if mode == 'f32' else 8e-3, 'conv_bn output %s %s' % (shape, mode))
    assert float(part.abs().max()) == 0.0
    yh = host(y).astype(np.float64).reshape(M, Cout)
    m_ref, v_ref = yh.mean(0), yh.var(0)
    scale = np.sqrt(v_ref).max()
    assert np.abs(host(mean) - m_ref).max() <= 2e-5 * scale + 1e-6
    assert np.abs(host(var) - v_ref).max() <= 1e-4 * v_ref.max()
    np.testing.assert_allclose(host(mm), mm0 - (mm0 - host(mean)) * np.float32(1 - 0.999), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(host(mv), mv0 - (mv0 - host(var)) * np.float32(1 - 0.999), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('wgs,shape,c32', [(3, (3, 40, 33, 32, 64, 3), True),      # 4182 positions on 3 workgroups: three tiles each
                                            (1, (4, 40, 33, 32, 64, 3), True),       # ... 5576 on one workgroup: 11 tiles, the ring wraps five times
                                            (2, (2, 12, 20, 32, 64, 3), True),       # 2 tiles, one per workgroup, the second range short
                                            (2, (1, 5, 300, 32, 64, 3), True),       # 300-wide rows: 144 KB of LDS
                                            (0, (1, 3, 500, 32, 64, 3), False)])     # a 500-wide row: the ring does not fit LDS -> the generic kernel
def test_conv_c32_persistent_tiles_and_width_fallback(ops, wgs, shape, c32):
    """conv_c32.hip gives every workgroup one contiguous range of the padded index and keeps one ring of pixel rows over it; with the workgroup
    count forced down every workgroup takes several tiles (ring wrap, the per-lane statistics carried across tiles).  Output, bias + leaky and the
    fused batch-norm sums against the oracle; the plan word says which kernel ran."""
    B, H, W, Cin, Cout, k = shape
    rng = np.random.RandomState(sum(shape) + 11)
    x = bf16_round(rng.randn(B, H, W, Cin).astype(np.float32))
    w = bf16_round((rng.randn(k, k, Cin, Cout) / np.sqrt(k * k * Cin)).astype(np.float32) + 0.02)
    bias = rng.randn(Cout).astype(np.float32)
    T = torch.bfloat16
    M = B * H * W
    F = torch.zeros(Cout * k * k * Cin, dtype=T, device='cuda')
    ops.filter_prep(dev(w), F, None, k, Cin, Cin, Cout, Cout, T)
    xd = dev(x, T)
    ws = torch.full((1024 + 256 * 256 * 128,), 3.0, dtype=torch.float32, device='cuda')
    ref = R.conv2d(x, w)
    try:
        if wgs:
            ops.set_stream_workgroups(wgs)
        y = torch.zeros(M * Cout, dtype=T, device='cuda')
        ops.conv2d_ws(xd, F, dev(bias), y, ws, B, H, W, Cin, Cin, Cout, Cout, k)
        plan = ops.last_conv_plan()
        assert ((plan['BM'], plan['BN'], plan['stages']) == (512, 64, 9)) == c32, plan
        yl = torch.zeros(M * Cout, dtype=T, device='cuda')
        ops.conv2d_bias_leaky(xd, F, dev(bias), yl, ws, B, H, W, Cin, Cin, Cout, Cout, k, 0.1)      # the inference epilogue (folded BN): bias, leaky_relu
        yb = torch.zeros(M * Cout, dtype=T, device='cuda')
        part = torch.zeros(2 * 256 * Cout, dtype=torch.float32, device='cuda')
        shift = dev((rng.randn(Cout) * 0.1).astype(np.float32))
        mean, var = torch.zeros(Cout, device='cuda'), torch.zeros(Cout, device='cuda')
        ops.conv2d_bn(xd, F, yb, ws, B, H, W, Cin, Cin, Cout, Cout, k, shift, part)
        planb = ops.last_conv_plan()
        assert ((planb['BM'], planb['BN'], planb['stages']) == (512, 64, 9)) == c32, planb
        ops.bn_finalize(part, shift, M, Cout, mean, var, None, None, 0.999)
        torch.cuda.synchronize()
    finally:
        ops.set_stream_workgroups(0)
    assert_close(host(y).reshape(B, H, W, Cout), ref + bias, BF16_RTOL, 'c32 fwd + bias %s wgs %d' % (shape, wgs))
    assert_close(host(yb).reshape(B, H, W, Cout), ref, BF16_RTOL, 'c32 fwd (bn) %s wgs %d' % (shape, wgs))
    z = ref + bias
    assert_close(host(yl).reshape(B, H, W, Cout), np.maximum(z, np.float32(0.1) * z), BF16_RTOL, 'c32 fwd + bias + leaky %s wgs %d' % (shape, wgs))
    y64 = host(yb).astype(np.float64).reshape(M, Cout)
    assert np.abs(host(mean) - y64.mean(0)).max() <= 2e-5 * np.sqrt(y64.var(0)).max() + 1e-6
    assert np.abs(host(var) - y64.var(0)).max() <= 1e-4 * y64.var(0).max()
    assert float(part.abs().max()) == 0.0


@pytest.mark.parametrize('wgs,shape,c64', [(3, (3, 40, 33, 64, 128, 3), True),     # 4182 positions on 3 workgroups: six tiles each, the ring wraps
                                            (1, (3, 40, 33, 64, 128, 3), True),      # ... on one workgroup: 17 tiles, the ring wraps five times
                                            (2, (2, 12, 20, 64, 128, 3), True),      # 546 positions on 2 workgroups: ranges of 384, the last tile of each partial
                                            (5, (2, 31, 16, 64, 128, 3), True),      # ranges of 256 positions: one tile per workgroup, the last range short
                                            (0, (2, 26, 27, 64, 128, 3), True),      # the product grid (one 128-position range per workgroup here)
                                            (2, (1, 6, 200, 64, 128, 3), True),      # 200-wide rows: 84 first pieces + a filter round = 157 KB of LDS
                                            (0, (1, 4, 320, 64, 128, 3), False)])    # a 320-wide row: first tile + filter round do not fit LDS -> the generic kernel
def test_conv_c64_persistent_tiles_and_width_fallback(ops, wgs, shape, c64):
    """conv_c64.hip (conv2 / conv4 forward: filter in registers, persistent workgroups over a contiguous range of the padded index, one ring of
    pixel rows).  With the workgroup count forced down every workgroup takes several tiles (ring wrap, partial last tiles, per-lane statistics
    carried across tiles).  Output, bias + leaky and the fused batch-norm sums against the oracle; the plan word says which kernel ran."""
    B, H, W, Cin, Cout, k = shape
    rng = np.random.RandomState(sum(shape) + 13)
    x = bf16_round(rng.randn(B, H, W, Cin).astype(np.float32))
    w = bf16_round((rng.randn(k, k, Cin, Cout) / np.sqrt(k * k * Cin)).astype(np.float32) + 0.02)
    bias = rng.randn(Cout).astype(np.float32)
    T = torch.bfloat16
    M = B * H * W
    F = torch.zeros(Cout * k * k * Cin, dtype=T, device='cuda')
    ops.filter_prep(dev(w), F, None, k, Cin, Cin, Cout, Cout, T)
    xd = dev(x, T)
    ws = torch.full((1024 + 256 * 256 * 128,), 3.0, dtype=torch.float32, device='cuda')
    ref = R.conv2d(x, w)
    try:
        if wgs:
            ops.set_stream_workgroups(wgs)
        y = torch.zeros(M * Cout, dtype=T, device='cuda')
        ops.conv2d_ws(xd, F, None, y, ws, B, H, W, Cin, Cin, Cout, Cout, k)
        plan = ops.last_conv_plan()
        assert ((plan['BM'], plan['BN'], plan['stages']) == (256, 128, 9)) == c64, plan
        yl = torch.zeros(M * Cout, dtype=T, device='cuda')
        ops.conv2d_bias_leaky(xd, F, dev(bias), yl, ws, B, H, W, Cin, Cin, Cout, Cout, k, 0.1)      # the inference epilogue (folded BN): bias, leaky_relu
        yb = torch.zeros(M * Cout, dtype=T, device='cuda')
        part = torch.zeros(2 * 256 * Cout, dtype=torch.float32, device='cuda')
        shift = dev((rng.randn(Cout) * 0.1).astype(np.float32))
        mean, var = torch.zeros(Cout, device='cuda'), torch.zeros(Cout, device='cuda')
        ops.conv2d_bn(xd, F, yb, ws, B, H, W, Cin, Cin, Cout, Cout, k, shift, part)
        planb = ops.last_conv_plan()
        assert ((planb['BM'], planb['BN'], planb['stages']) == (256, 128, 9)) == c64, planb
        ops.bn_finalize(part, shift, M, Cout, mean, var, None, None, 0.999)
        torch.cuda.synchronize()
    finally:
        ops.set_stream_workgroups(0)
    assert_close(host(y).reshape(B, H, W, Cout), ref, BF16_RTOL, 'c64 fwd %s wgs %d' % (shape, wgs))
    assert_close(host(yb).reshape(B, H, W, Cout), ref, BF16_RTOL, 'c64 fwd (bn) %s wgs %d' % (shape, wgs))
    assert torch.equal(y, yb)                     # the statistics epilogue does not change what is stored
    z = ref + bias
    assert_close(host(yl).reshape(B, H, W, Cout), np.maximum(z, np.float32(0.1) * z), BF16_RTOL, 'c64 fwd + bias + leaky %s wgs %d' % (shape, wgs))
    # per element against the oracle on the bf16-rounded operands: one bf16 ulp of the stored output
    got = host(y).reshape(B, H, W, Cout).astype(np.float64)
    assert (np.abs(got - ref) <= 4e-3 * np.abs(ref) + 2e-4 * np.abs(ref).max()).all()
    y64 = host(yb).astype(np.float64).reshape(M, Cout)
    assert np.abs(host(mean) - y64.mean(0)).max() <= 2e-5 * np.sqrt(y64.var(0)).max() + 1e-6
    assert np.abs(host(var) - y64.var(0)).max() <= 1e-4 * y64.var(0).max()
    assert float(part.abs().max()) == 0.0


def _bn_bwd_sums_oracle(dA, yprev, mean, var, gamma, beta, eps, alpha=0.1):
    """dgamma, dbeta of a = leaky(bn(yprev)) for the activation gradient dA, by the ORACLE's formulas (R.bn_apply, R.leaky_relu_grad,
    R.bn_train_bwd); inputs [M, C] float arrays."""
    M, C = dA.shape
    y4 = yprev.reshape(1, 1, M, C).astype(np.float32)
    z = R.bn_apply(y4, mean, var, gamma, beta, eps)
    dz = R.leaky_relu_grad(z, dA.reshape(1, 1, M, C).astype(np.float32), alpha)
    _, dg, db = R.bn_train_bwd(y4, mean, var, gamma, dz, eps)
    return dg.astype(np.float64), db.astype(np.float64)


TAP_SHAPES = [(16, 13, 13, 512, 1024), (16, 26, 26, 256, 512), (9, 52, 52, 128, 256), (8, 13, 13, 1024, 504), (16, 26, 26, 256, 136),
              (16, 55, 55, 128, 128), (16, 13, 13, 3072, 1024), (16, 13, 13, 1024, 512), (16, 26, 26, 512, 256)]
# variant -> (yolo2_debug_set_igemm_tap mode, ping-pong grid (1 stream-K / 2 one workgroup per tile), ping-pong SCHED (-1 = the default))
# s4 / s4_tiles: the loader / consumer member of the family (conv_s4.hip: four computing waves of 128 x 64 + four loader waves; plan word waves == 4)
TAP_VARIANTS = {'pp': (2, 1, -1), 'pp_tiles': (2, 2, -1), 's4': (3, 1, -1), 's4_tiles': (3, 2, -1)}
_TAP_ORACLE = {}       # shape -> oracle convolution (the same for every variant and epilogue: minutes of CPU time when recomputed 12 times)


TAP_EPILOGUES = ['plain', 'bias_leaky', 'bn_stats', 'dgrad_bn']
# stream-K: every shape x every epilogue.  One workgroup per tile: the bench shapes the launch rule gives a tile grid (26x26 256->512 forward,
# 52x52 / 55x55 data gradients).
TAP_CASES = [(s_, v_, e_) for v_ in ('pp', 's4') for e_ in TAP_EPILOGUES for s_ in TAP_SHAPES] + \
            [(s_, v_, e_) for v_ in ('pp_tiles', 's4_tiles') for e_ in ('plain', 'bn_stats', 'dgrad_bn') for s_ in (TAP_SHAPES[1], TAP_SHAPES[2], TAP_SHAPES[5])]
# Shapes with fewer K steps than CUs (odd sizes, partial tiles, 3 and 5 chunks, H != W).  A forced stream-K grid used to leave workgroups without
# work whose flags an owner then waited for (a hang, found by these shapes); launch_conv clamps the grid to the number of K steps
# (green on hardware: profiles/r05_tiny_tap_shapes.txt).
_tiny = [(3, 5, 7, 64, 72), (2, 19, 19, 128, 200), (1, 27, 28, 192, 128), (5, 10, 10, 320, 264)]
TAP_CASES += [(s_, v_, e_) for e_ in TAP_EPILOGUES for v_ in ('pp', 'pp_tiles', 's4', 's4_tiles') for s_ in _tiny]
# batch-normalised outputs / producer activations are stored unpadded: those epilogues only with a filter count that is a multiple of 8
TAP_CASES = [c_ for c_ in TAP_CASES if c_[2] in ('plain', 'bias_leaky') or c_[0][4] % 8 == 0]


@pytest.mark.parametrize('shape,variant,epilogue', TAP_CASES, ids=['%s-%s-%s' % ('x'.join(map(str, c[0])), c[1], c[2]) for c in TAP_CASES])
def test_conv_tap_fused_3x3(ops, shape, variant, epilogue):
    """The ping-pong tap-fused 3x3 kernel (conv_pp.hip: one halo image per 64-channel chunk, nine taps read from it), as stream-K and with
    one workgroup per tile, against the per-tap kernel on the same operands -- same products, a different f32 summation order across
    stream-K segments only -- and against the oracle."""
    B, H, W, Cin, Cout = shape
    mode, pp_grid, pp_dma = TAP_VARIANTS[variant]
    k, M = 3, B * H * W
    rng = np.random.RandomState(sum(shape) + 3)
    x = bf16_round(rng.randn(B, H, W, Cin).astype(np.float32))
    w = bf16_round((rng.randn(k, k, Cin, Cout) / np.sqrt(k * k * Cin)).astype(np.float32))
    T = torch.bfloat16
    ldo = ops.pad8(Cout)
    F = torch.zeros(Cout * k * k * Cin, dtype=T, device='cuda')
    ops.filter_prep(dev(w), F, None, k, Cin, Cin, Cout, ldo, T)
    xd = dev(x, T)
    ws = torch.full((1024 + 256 * 256 * 128,), 3.0, dtype=torch.float32, device='cuda')
    bias = dev(rng.randn(Cout).astype(np.float32))
    out = {}
    for tap in (0, 1):
        ops.set_igemm_tap(mode if tap else 0)
        ops.set_pp(grid=pp_grid, dmapos=pp_dma, min_steps=0, min_share=0)      # every shape of this test takes the ping-pong kernel (dmapos < 0: keep the default SCHED)
        try:
            O = torch.zeros(M * ldo, dtype=T, device='cuda')
            extra = None
            if epilogue == 'plain':
                ops.conv2d_ws(xd, F, None, O, ws, B, H, W, Cin, Cin, Cout, ldo, k)
            elif epilogue == 'bias_leaky':
                ops.conv2d_bias_leaky(xd, F, bias, O, ws, B, H, W, Cin, Cin, Cout, ldo, k, 0.1)
            elif epilogue == 'bn_stats':
                assert ldo == Cout          # (batch-normalised outputs are stored unpadded: TAP_CASES pairs this epilogue with such shapes only)
                part = torch.zeros(2 * 256 * Cout, dtype=torch.float32, device='cuda')
                shift = dev(rng.randn(Cout).astype(np.float32) * 0.05) if tap == 0 else out[0][2][3]
                mean, var = torch.zeros(Cout, device='cuda'), torch.zeros(Cout, device='cuda')
                ops.conv2d_bn(xd, F, O, ws, B, H, W, Cin, Cin, Cout, Cout, k, shift, part)
                ops.bn_finalize(part, shift, M, Cout, mean, var, None, None, 0.999)
                extra = (mean, var, part, shift)
            else:
                assert ldo == Cout
                if tap == 0:
                    yprev = dev(rng.randn(M, Cout).astype(np.float32) * 1.5 + 0.3, T)
                    pm, pv = dev(rng.randn(Cout).astype(np.float32) * 0.2 + 0.3), dev((rng.rand(Cout) + 0.5).astype(np.float32))
                    pg, pb = dev((rng.rand(Cout) + 0.5).astype(np.float32)), dev(rng.randn(Cout).astype(np.float32) * 0.3)
                else:
                    yprev, pm, pv, pg, pb = out[0][2][4:]
                part = torch.zeros(2 * 256 * Cout, dtype=torch.float32, device='cuda')
                red = torch.zeros(ops.workspace_bytes('bn', Cout) // 8, dtype=torch.float64, device='cuda')
                dg, db = torch.zeros(Cout, device='cuda'), torch.zeros(Cout, device='cuda')
                if ops.conv2d_dgrad_bn(xd, F, O, ws, B, H, W, Cin, Cin, Cout, ldo, k, yprev, pm, pv, pg, pb, dg, db, part, red, 1e-3, 0.1):
                    ops.bn_part_to_grads(part, Cout, dg, db)
                extra = (dg, db, part, None, yprev, pm, pv, pg, pb)
            plan = ops.last_conv_plan()
            torch.cuda.synchronize()
            assert (plan['stages'] == 18) == bool(tap and Cout > 64), plan
            if tap and Cout > 64:
                assert plan['waves'] == (4 if mode == 3 else 8), plan
            if tap and pp_grid == 2:
                assert plan['grid_x'] == -(-M // 256) * -(-Cout // 128), plan
            out[tap] = (host(O).reshape(M, ldo), plan, extra)
        finally:
            ops.set_igemm_tap(2)
            ops.set_pp(grid=0, min_steps=18, min_share=24)
    y0, y1 = out[0][0], out[1][0]
    assert np.all(y1[:, Cout:] == 0)
    assert_close(y1, y0, 8e-3, 'tap-fused vs per-tap %s %s' % (shape, epilogue))     # one bf16 ulp where the f32 sums round differently
    assert np.mean(y1 != y0) < 0.02
    if epilogue in ('plain', 'bias_leaky'):      # (every shape, the 3072-channel one included: the oracle result is cached per shape)
        if shape not in _TAP_ORACLE:
            _TAP_ORACLE[shape] = R.conv2d(x, w).reshape(M, Cout)
        ref = _TAP_ORACLE[shape]
        if epilogue == 'bias_leaky':
            ref = ref + host(bias)
            ref = np.maximum(ref, 0.1 * ref)
        assert_close(y1[:, :Cout], ref, BF16_RTOL, 'tap-fused vs oracle %s %s' % (shape, epilogue))
    if epilogue == 'bn_stats':
        for a, b, name in ((out[1][2][0], out[0][2][0], 'mean'), (out[1][2][1], out[0][2][1], 'var')):
            assert np.abs(host(a) - host(b)).max() <= 2e-3 * np.abs(host(b)).max() + 1e-5, name
        assert float(out[1][2][2].abs().max()) == 0.0
        yh = y1.astype(np.float64)
        assert np.abs(host(out[1][2][0]) - yh.mean(0)).max() <= 2e-5 * np.sqrt(yh.var(0)).max() + 1e-6
        assert np.abs(host(out[1][2][1]) - yh.var(0)).max() <= 1e-4 * yh.var(0).max()
    if epilogue == 'dgrad_bn':
        for a, b, name in ((out[1][2][0], out[0][2][0], 'dgamma'), (out[1][2][1], out[0][2][1], 'dbeta')):
            assert np.abs(host(a) - host(b)).max() <= 2e-2 * np.abs(host(b)).max(), name
        assert float(out[1][2][2].abs().max()) == 0.0
        # ... and DIRECTLY against the oracle's BN + leaky backward sums, evaluated on the gradient tile the kernel stored (y1) --
        # the fused epilogue is not only "equal to the unfused GPU pair"
        _, _, _, _, yprev, pm, pv, pg, pb = out[1][2]
        dg_o, db_o = _bn_bwd_sums_oracle(y1[:, :Cout], host(yprev).reshape(M, Cout), host(pm), host(pv), host(pg), host(pb), 1e-3)
        for got, ref, name in ((out[1][2][0], dg_o, 'dgamma'), (out[1][2][1], db_o, 'dbeta')):
            assert np.abs(host(got) - ref).max() <= 3e-3 * np.abs(ref).max(), (name, np.abs(host(got) - ref).max(), np.abs(ref).max())


def test_streamk_unserviceable_grid_surfaces_as_error(ops):
    """A stream-K partition with workgroups that hold no K step (forced: 16 workgroups on 9 steps, the grid clamp of launch_conv switched
    off) leaves flags nobody raises.  The owner's wait is bounded (conv_shared.h y2_sk_wait_and_clear): the launch ends, and
    yolo2_check_async_errors reports it as YOLO2_E_LAUNCH -- errors surface, the device never hangs (the reference's failures are
    exceptions too: utils/postprocess.py:22-35, detect.py:70).  Afterwards the library works again."""
    B, H, W, Cin, Cout, k = 3, 5, 7, 64, 72, 3
    M = B * H * W
    rng = np.random.RandomState(77)
    x = bf16_round(rng.randn(B, H, W, Cin).astype(np.float32))
    w = bf16_round((rng.randn(k, k, Cin, Cout) / np.sqrt(k * k * Cin)).astype(np.float32))
    T = torch.bfloat16
    F = torch.zeros(Cout * k * k * Cin, dtype=T, device='cuda')
    ops.filter_prep(dev(w), F, None, k, Cin, Cin, Cout, Cout, T)
    xd = dev(x, T)
    ws = torch.full((1024 + 256 * 256 * 128,), 3.0, dtype=torch.float32, device='cuda')
    O = torch.zeros(M * Cout, dtype=T, device='cuda')
    ops.check_async_errors()                                   # clean before
    ops.set_streamk_wait_us(2000, unclamped=True)              # 2 ms per unanswered flag instead of 2 s
    ops.set_pp(grid=16, min_steps=0, min_share=0)
    try:
        ops.conv2d_ws(xd, F, None, O, ws, B, H, W, Cin, Cin, Cout, Cout, k)
        plan = ops.last_conv_plan()
        assert plan['stages'] == 18 and plan['grid_x'] == 16, plan
        with pytest.raises(RuntimeError, match='stream-K hand-off'):
            ops.check_async_errors()
    finally:
        ops.set_pp(grid=0, min_steps=18, min_share=24)
        ops.set_streamk_wait_us(0, unclamped=False)
    ops.check_async_errors()                                   # the status was consumed, the pool reset
    ops.set_pp(grid=1, min_steps=0, min_share=0)               # the same launch with the clamp: 9 workgroups, correct
    try:
        ops.conv2d_ws(xd, F, None, O, ws, B, H, W, Cin, Cin, Cout, Cout, k)
        assert ops.last_conv_plan()['grid_x'] == 9
        ops.check_async_errors()
    finally:
        ops.set_pp(grid=0, min_steps=18, min_share=24)
    assert_close(host(O).reshape(M, Cout), R.conv2d(x, w).reshape(M, Cout), BF16_RTOL, 'stream-K after a reported hand-off failure')


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
@pytest.mark.parametrize('shape,fused', [((2, 26, 26, 160, 64, 3), True),       # (B, H, W, filters of the consumer, channels = producer filters, k); M tail
                                         ((16, 13, 13, 1024, 512, 3), True),   # stream-K (256 x 128 tiles): owners hold the finished tiles
                                         ((8, 13, 13, 1024, 504, 3), True),    # stream-K, ragged filter tile
                                         ((4, 52, 52, 256, 128, 3), True),     # full grid, wide rows
                                         ((2, 26, 26, 512, 256, 1), True),
                                         ((3, 20, 20, 64, 32, 3), True),       # 32-wide filter tile
                                         ((2, 13, 13, 512, 256, 3), False),   # K-sliced grid -> two-step form inside the call
                                         ((2, 104, 104, 64, 128, 1), True),    # conv3's data gradient: the persistent 1x1 kernel (conv_d1.hip, bf16), 64-channel reduction
                                         ((4, 52, 52, 128, 256, 1), True),     # conv6's: 128-channel reduction, two column groups
                                         ((3, 53, 55, 64, 384, 1), True)])     # ragged last tile, three column groups, more tiles than workgroups leave whole
def test_conv_dgrad_bn_fused_sums(ops, shape, fused, mode):
    """yolo2_conv2d_dgrad_bn == yolo2_conv2d_ws followed by yolo2_bn_leaky_bwd_reduce: the same dX bit for bit, dgamma / dbeta equal up
    to the f32 summation order; the partial rows are zero again afterwards."""
    B, H, W, Cout, Cin, k = shape
    tdtype = torch.float32 if mode == 'f32' else torch.bfloat16
    rng = np.random.RandomState(sum(shape) + 11)
    M = B * H * W
    ldy, ldx = ops.pad8(Cout), ops.pad8(Cin)
    dy = dev(pad_channels(rng.randn(B, H, W, Cout).astype(np.float32), ldy), tdtype)
    w = (rng.randn(k, k, Cin, Cout) / np.sqrt(k * k * Cout)).astype(np.float32)
    F = torch.zeros(Cin * k * k * ldy, dtype=tdtype, device='cuda')
    ops.filter_prep(dev(w), None, F, k, Cin, ldx, Cout, ldy, tdtype)
    yprev = dev(rng.randn(M, Cin).astype(np.float32) * 1.5 + 0.3, tdtype)
    mean, var = dev(rng.randn(Cin).astype(np.float32) * 0.2 + 0.3), dev((rng.rand(Cin) + 0.5).astype(np.float32))
    gamma, beta = dev((rng.rand(Cin) + 0.5).astype(np.float32)), dev(rng.randn(Cin).astype(np.float32) * 0.3)
    ws = torch.full((1024 + 256 * 256 * 128,), 3.0, dtype=torch.float32, device='cuda')
    red = torch.zeros(ops.workspace_bytes('bn', Cin) // 8, dtype=torch.float64, device='cuda')
    dx_ref = torch.zeros(M * ldx, dtype=tdtype, device='cuda')
    ops.conv2d_ws(dy, F, None, dx_ref, ws, B, H, W, ldy, ldy, Cin, ldx, k)
    dg_ref, db_ref = torch.zeros(Cin, device='cuda'), torch.zeros(Cin, device='cuda')
    ops.bn_leaky_bwd_reduce(dx_ref, ldx, yprev, mean, var, gamma, beta, dg_ref, db_ref, red, M, Cin, 1e-3, 0.1)
    dx = torch.zeros(M * ldx, dtype=tdtype, device='cuda')
    dg, db = torch.full((Cin,), 9.0, device='cuda'), torch.full((Cin,), 9.0, device='cuda')
    part = torch.zeros(2 * 256 * Cin, dtype=torch.float32, device='cuda')
    pending = ops.conv2d_dgrad_bn(dy, F, dx, ws, B, H, W, ldy, ldy, Cin, ldx, k, yprev, mean, var, gamma, beta, dg, db, part, red, 1e-3, 0.1)
    plan = ops.last_conv_plan()
    assert pending == bool(plan['split'] & 0x100), (pending, plan)
    if mode == 'bf16' and k == 1 and Cout in (64, 128) and Cin % 128 == 0 and M >= 8192 and os.environ.get('YOLO2_D1', '1') != '0':
        assert (plan['BM'], plan['stages'], plan['grid_y']) == (128, 1, Cin // 128) and plan['grid_x'] <= 32, plan      # conv_d1.hip: persistent, 32 partial rows
    if mode == 'bf16':                     # (f32 tiles of some variants do not fit the LDS image: those run the two-step form)
        assert pending == (fused and os.environ.get('YOLO2_FUSE_BN_BWD', '1') != '0'), (pending, plan)
    if pending:
        ops.bn_part_to_grads(part, Cin, dg, db)
    torch.cuda.synchronize()
    # (stream-K / K-sliced grids add partial tiles in a launch-dependent order: last-bit differences in dX between launches)
    assert_close(host(dx), host(dx_ref), 1e-6 if mode == 'f32' else 8e-3, 'dgrad_bn dX %s %s' % (shape, mode))
    assert float(part.abs().max()) == 0.0
    if torch.equal(dx, dx_ref):
        tol = 2e-5
    else:                                  # a differently rounded dX element moves the sums by more than their summation-order noise
        tol = 2e-4 if mode == 'f32' else 2e-2
    for got, ref, name in ((dg, dg_ref, 'dgamma'), (db, db_ref, 'dbeta')):
        g, r = host(got).astype(np.float64), host(ref).astype(np.float64)
        assert np.abs(g - r).max() <= tol * np.abs(r).max(), (name, shape, mode, np.abs(g - r).max(), np.abs(r).max())
    # the same three results DIRECTLY against the oracle (the comparison above is GPU vs GPU): dX vs the oracle's data gradient on the
    # same (rounded) operands, dgamma / dbeta vs the oracle's BN + leaky backward sums of the dX the kernel stored
    rnd = (lambda a: a) if mode == 'f32' else bf16_round
    dy_np = host(dy).reshape(B, H, W, ldy)[..., :Cout]
    dx_o = R.conv2d_dgrad(dy_np, rnd(w)).reshape(M, Cin)
    dx_g = host(dx).reshape(M, ldx)[:, :Cin]
    assert_close(dx_g, dx_o, F32_RTOL if mode == 'f32' else BF16_RTOL, 'dgrad_bn dX vs oracle %s %s' % (shape, mode))
    dg_o, db_o = _bn_bwd_sums_oracle(dx_g, host(yprev).reshape(M, Cin), host(mean), host(var), host(gamma), host(beta), 1e-3)
    for got, ref, name in ((dg, dg_o, 'dgamma'), (db, db_o, 'dbeta')):
        assert np.abs(host(got) - ref).max() <= (3e-4 if mode == 'f32' else 3e-3) * np.abs(ref).max(), (name, shape, mode)


WGRAD_SHAPES = CONV_SHAPES + [(2, 13, 13, 256, 128, 3), (2, 26, 26, 128, 256, 1), (4, 52, 52, 32, 64, 3)]


def test_conv_wgrad_single_range_overwrites(ops):
    """Shapes whose filter gradient is one pixel range per tile store instead of accumulating: dW may be dirty
    (yolo2_conv2d_wgrad_accumulates == 0); split shapes report 1."""
    B, H, W, Cin, Cout, k = 2, 13, 13, 1024, 1024, 3          # 576 tiles -> single range
    assert not ops.conv2d_wgrad_accumulates(B, H, W, Cin, Cin, Cout, Cout, k, torch.bfloat16)
    assert ops.conv2d_wgrad_accumulates(16, 104, 104, 64, 64, 128, 128, 3, torch.bfloat16)
    assert ops.conv2d_wgrad_accumulates(16, 416, 416, 3, 8, 32, 32, 3, torch.bfloat16)
    rng = np.random.RandomState(5)
    x = bf16_round(rng.randn(B, H, W, Cin).astype(np.float32))
    dy = bf16_round(rng.randn(B, H, W, Cout).astype(np.float32))
    ref = R.conv2d_wgrad(x, dy, k, k)
    dW = torch.full((k * k * Cin * Cout,), 123.0, dtype=torch.float32, device='cuda')     # dirty on purpose
    ops.conv2d_wgrad(dev(x, torch.bfloat16), dev(dy, torch.bfloat16), dW, B, H, W, Cin, Cin, Cout, Cout, k)
    torch.cuda.synchronize()
    assert_close(host(dW).reshape(k, k, Cin, Cout), ref, BF16_RTOL, 'wgrad single range, dirty dW')


# bf16: the product rule (3x3: the row-of-taps kernel of conv_wgrad3.hip, else the per-tap kernel); bf16_gather: scalar reference gather;
# bf16_pertap: the per-tap transpose-read kernel for every shape; bf16_row2 / 4 / 5: each product variant of the row-of-taps kernel forced
WGRAD_MODES = {'f32': 0, 'bf16': 0, 'bf16_gather': 1, 'bf16_pertap': 2, 'bf16_row2': 12, 'bf16_row4': 14, 'bf16_row5': 15}
WGRAD_SHAPES += [(2, 13, 13, 96, 200, 3),     # ragged channel and filter tiles of every row-kernel variant
                 (1, 16, 15, 64, 64, 3),      # W + 1 and H powers of two (the division constants' special case)
                 (3, 9, 31, 40, 72, 3)]


# (the row-of-taps kernel takes 3x3 layers beyond the image layer: its forced variants are only paired with those shapes)
WGRAD_CASES = [(s_, m_) for m_ in WGRAD_MODES for s_ in WGRAD_SHAPES if not (m_.startswith('bf16_row') and (s_[5] != 3 or s_[3] <= 8))]


@pytest.mark.parametrize('shape,mode', WGRAD_CASES, ids=['%s-%s' % ('x'.join(map(str, c[0])), c[1]) for c in WGRAD_CASES])
def test_conv_wgrad(ops, shape, mode):
    B, H, W, Cin, Cout, k = shape
    tdtype = torch.float32 if mode == 'f32' else torch.bfloat16
    rng = np.random.RandomState(sum(shape) + 2)
    x = rng.randn(B, H, W, Cin).astype(np.float32)
    dy = rng.randn(B, H, W, Cout).astype(np.float32)
    if mode != 'f32':
        x, dy = bf16_round(x), bf16_round(dy)
    # bf16 operands are exact, their products exact in f32: the f64 oracle is THE answer and the kernel's only error is its f32 summation order
    ref = R.conv2d_wgrad(x, dy, k, k) if mode == 'f32' else R.conv2d_wgrad(x.astype(np.float64), dy.astype(np.float64), k, k)
    ldx, ldy = ops.pad8(Cin), ops.pad8(Cout)
    dW = torch.zeros(k * k * Cin * Cout, dtype=torch.float32, device='cuda')
    ops.set_wgrad_variant(WGRAD_MODES[mode])
    try:
        ops.conv2d_wgrad(dev(pad_channels(x, ldx), tdtype), dev(pad_channels(dy, ldy), tdtype), dW, B, H, W, Cin, ldx, Cout, ldy, k)
        torch.cuda.synchronize()
        plan = ops.last_wgrad_plan()
    finally:
        ops.set_wgrad_variant(0)
    if mode.startswith('bf16_row'):
        assert plan['pair'] == 3, plan          # three taps per workgroup: the row kernel ran
    got = host(dW).reshape(k, k, Cin, Cout)
    if mode == 'f32':
        assert_close(got, ref, F32_RTOL, 'conv wgrad %s %s' % (shape, mode))
    else:
        assert_wgrad_exact_products(got, ref, 'conv wgrad %s %s' % (shape, mode))


def _border_mask(H, W, which):
    m = np.zeros((H, W), bool)
    if which in ('first_row', 'frame'): m[0, :] = True
    if which in ('last_row', 'frame'): m[H - 1, :] = True
    if which in ('first_col', 'frame'): m[:, 0] = True
    if which in ('last_col', 'frame'): m[:, W - 1] = True
    if which == 'corners': m[0, 0] = m[0, W - 1] = m[H - 1, 0] = m[H - 1, W - 1] = True
    return m


BORDERS = ['first_row', 'last_row', 'first_col', 'last_col', 'corners', 'frame']


@pytest.mark.parametrize('which', BORDERS)
@pytest.mark.parametrize('mode', ['bf16', 'bf16_row2', 'bf16_row4', 'bf16_row5', 'bf16_pertap'])
@pytest.mark.parametrize('shape', [(2, 13, 13, 128, 128), (2, 16, 15, 64, 64), (1, 26, 27, 64, 128), (3, 7, 52, 128, 64), (3, 4, 32, 32, 64)])
def test_conv_wgrad_border_only_inputs(ops, shape, mode, which):
    """Padded-index kernels fail at image borders first (conv_wgrad3.hip: the left neighbour of column 0 is the previous row's zero column,
    rows outside the image read as zeros through the buffer range check).  Inputs that are non-zero ONLY on the border -- in x, then in dy --
    make every surviving term of dW a border term: a wrapped column, a row taken from the neighbouring image or a dropped edge pixel changes
    whole elements, not the sixth digit.  Per element against the f64 sum of the exact products."""
    B, H, W, Cin, Cout = shape
    rng = np.random.RandomState(sum(shape) + len(which))
    mask = _border_mask(H, W, which)[None, :, :, None]
    T = torch.bfloat16
    for side in ('x', 'dy'):
        x = bf16_round(rng.randn(B, H, W, Cin).astype(np.float32))
        dy = bf16_round(rng.randn(B, H, W, Cout).astype(np.float32))
        if side == 'x': x = x * mask
        else: dy = dy * mask
        ref = R.conv2d_wgrad(x.astype(np.float64), dy.astype(np.float64), 3, 3)
        dW = torch.zeros(9 * Cin * Cout, dtype=torch.float32, device='cuda')
        ops.set_wgrad_variant(WGRAD_MODES[mode])
        try:
            ops.conv2d_wgrad(dev(x, T), dev(dy, T), dW, B, H, W, Cin, Cin, Cout, Cout, 3)
            torch.cuda.synchronize()
            plan = ops.last_wgrad_plan()
        finally:
            ops.set_wgrad_variant(0)
        if mode.startswith('bf16_row'):
            assert plan['pair'] == 3, plan
        assert_wgrad_exact_products(host(dW).reshape(3, 3, Cin, Cout), ref, 'wgrad border-only %s in %s %s %s' % (which, side, shape, mode))


C32W_SHAPES = [(2, 16, 16), (3, 5, 32), (1, 9, 208), (2, 31, 48), (1, 1, 16), (1, 2, 16), (1, 3, 304), (5, 4, 224), (2, 2, 240), (7, 3, 64)]


@pytest.mark.parametrize('shape', C32W_SHAPES, ids=['x'.join(map(str, c)) for c in C32W_SHAPES])
def test_conv1_wgrad_all_taps_per_workgroup(ops, shape):
    """conv_wgrad_c32.hip: the filter gradient of a 32 -> 64 channel 3x3 layer (Darknet-19 conv1; reference model/yolo2/inference.py:76, gradient by
    train.py:127-129) with all nine taps in one workgroup over a run of image rows: X rows R - 1 / R / R + 1 from a ring of image rows, a kernel row
    that leaves the image sits the step out.  Shapes: rows that end images inside a workgroup's run (B > 1 with few rows), one-row images (both
    vertical neighbours outside), one workgroup (plain stores into a dirty buffer), 224 / 240 / 304-wide rows (the deepest / the shallower ring).
    Per element against the f64 sum of the exact bf16 products."""
    B, H, W = shape
    Cin, Cout, T = 32, 64, torch.bfloat16
    rng = np.random.RandomState(sum(shape) + 5)
    x = bf16_round(rng.randn(B, H, W, Cin).astype(np.float32))
    dy = bf16_round(rng.randn(B, H, W, Cout).astype(np.float32))
    ref = R.conv2d_wgrad(x.astype(np.float64), dy.astype(np.float64), 3, 3)
    accumulates = ops.conv2d_wgrad_accumulates(B, H, W, Cin, Cin, Cout, Cout, 3, T)
    dW = torch.zeros(9 * Cin * Cout, dtype=torch.float32, device='cuda') if accumulates else torch.full((9 * Cin * Cout,), 7.0, dtype=torch.float32, device='cuda')
    ops.conv2d_wgrad(dev(x, T), dev(dy, T), dW, B, H, W, Cin, Cin, Cout, Cout, 3)
    torch.cuda.synchronize()
    plan = ops.last_wgrad_plan()
    assert (plan['BC'], plan['BN'], plan['waves'], plan['pair']) == (32, 64, 12, 9), plan
    assert bool(plan['direct']) == (not accumulates) and (plan['blocks'] == 1) == (not accumulates), (plan, accumulates)
    assert_wgrad_exact_products(host(dW).reshape(3, 3, Cin, Cout), ref, 'conv1 wgrad, nine taps per workgroup %s' % (shape,))


@pytest.mark.parametrize('chan', [(32, 64, 256), (64, 128, 128)], ids=['conv1', 'conv2_4'])      # (Cin, Cout of the LAYER, plan BM of its data gradient)
@pytest.mark.parametrize('wgs,shape', [(3, (3, 40, 33)), (1, (2, 31, 16)), (2, (1, 9, 208)), (0, (2, 26, 27))])
def test_conv_c64_32_filter_form_as_data_gradient(ops, wgs, shape, chan):
    """conv_c64.hip with 32 filters: the data gradient of a 32 -> 64 channel 3x3 layer (Darknet-19 conv1; reference model/yolo2/inference.py:76) is
    the forward convolution of dY (64 channels) with the flipped, transposed filters.  Every wave holds the whole operand and takes 32 positions of
    a tile; accumulators alternate over the K steps.  The same for a 64 -> 128 channel layer (conv2 / conv4): 128-channel dY, 64 filters -- the waves of a
    pair take one channel half each and swap partial blocks through LDS (conv_c64_wide_kernel).  Against the oracle's conv2d_dgrad; with a bias /
    activation the call takes the generic kernel."""
    B, H, W = shape
    Cin, Cout, k = chan[0], chan[1], 3
    if W > 150 and Cout == 128:
        pytest.skip('208-wide rows of 128 channels do not fit the two-plane ring (the layer runs at 104 x 104)')
    rng = np.random.RandomState(sum(shape) + 77)
    dy = bf16_round(rng.randn(B, H, W, Cout).astype(np.float32))
    w = bf16_round((rng.randn(k, k, Cin, Cout) / np.sqrt(k * k * Cout)).astype(np.float32))
    T = torch.bfloat16
    Fd = torch.zeros(Cin * k * k * Cout, dtype=T, device='cuda')
    ops.filter_prep(dev(w), None, Fd, k, Cin, Cin, Cout, Cout, T)
    ws = torch.full((1024 + 256 * 256 * 128,), 3.0, dtype=torch.float32, device='cuda')
    dx = torch.full((B * H * W * Cin,), 5.0, dtype=T, device='cuda')
    dxl = torch.zeros(B * H * W * Cin, dtype=T, device='cuda')
    try:
        if wgs:
            ops.set_stream_workgroups(wgs)
        ops.conv2d_ws(dev(dy, T), Fd, None, dx, ws, B, H, W, Cout, Cout, Cin, Cin, k)
        plan = ops.last_conv_plan()
        ops.conv2d_bias_leaky(dev(dy, T), Fd, dev(np.zeros(Cin, np.float32)), dxl, ws, B, H, W, Cout, Cout, Cin, Cin, k, 0.1)
        planl = ops.last_conv_plan()
        torch.cuda.synchronize()
    finally:
        ops.set_stream_workgroups(0)
    assert (plan['BM'], plan['BN'], plan['stages']) == (chan[2], Cin, 9), plan
    assert (planl['BM'], planl['BN'], planl['stages']) != (chan[2], Cin, 9), planl
    ref = R.conv2d_dgrad(dy.astype(np.float64), w.astype(np.float64))
    got = host(dx).reshape(B, H, W, Cin).astype(np.float64)
    assert (np.abs(got - ref) <= 4e-3 * np.abs(ref) + 2e-4 * np.abs(ref).max()).all(), 'c64 32-filter dgrad %s wgs %d: %.3e' % (shape, wgs, np.abs(got - ref).max())
    assert_close(host(dxl).reshape(B, H, W, Cin), np.maximum(ref, 0.1 * ref), BF16_RTOL, 'generic kernel on the same operands')


@pytest.mark.parametrize('which', BORDERS)
@pytest.mark.parametrize('chan', [(32, 64, 512, 64), (64, 128, 256, 128), (64, 32, 256, 32), (128, 64, 128, 64)], ids=['c32', 'c64', 'c64n', 'c64w'])      # (Cin, Cout, plan BM, plan BN): conv_c32.hip / conv_c64.hip (128 and 32 filters)
@pytest.mark.parametrize('shape', [(2, 12, 20), (1, 7, 9), (3, 40, 33), (2, 31, 16)])
def test_conv_c32_border_only_inputs(ops, shape, which, chan):
    """conv_c32.hip (conv1 forward, padded position index with one zero column per row and one zero row per image): an input that is non-zero
    only on the image border, against the oracle; every output element further than one pixel from the border must be EXACTLY zero."""
    B, H, W = shape
    Cin, Cout, k = chan[0], chan[1], 3
    rng = np.random.RandomState(sum(shape) + len(which))
    mask = _border_mask(H, W, which)
    x = bf16_round(rng.randn(B, H, W, Cin).astype(np.float32)) * mask[None, :, :, None]
    w = bf16_round((rng.randn(k, k, Cin, Cout) / np.sqrt(k * k * Cin)).astype(np.float32))
    T = torch.bfloat16
    F = torch.zeros(Cout * k * k * Cin, dtype=T, device='cuda')
    ops.filter_prep(dev(w), F, None, k, Cin, Cin, Cout, Cout, T)
    ws = torch.full((1024 + 256 * 256 * 128,), 3.0, dtype=torch.float32, device='cuda')
    y = torch.full((B * H * W * Cout,), 7.0, dtype=T, device='cuda')
    ops.conv2d_ws(dev(x, T), F, None, y, ws, B, H, W, Cin, Cin, Cout, Cout, k)
    plan = ops.last_conv_plan()
    torch.cuda.synchronize()
    assert (plan['BM'], plan['BN'], plan['stages']) == (chan[2], chan[3], 9), plan
    got = host(y).reshape(B, H, W, Cout)
    ref = R.conv2d(x.astype(np.float64), w.astype(np.float64))
    scale = np.abs(ref).max()
    assert (np.abs(got - ref) <= 4e-3 * np.abs(ref) + 1e-6 * scale).all(), 'c32 border-only %s %s: max err %.3e' % (which, shape, np.abs(got - ref).max())
    # reach of the mask through a 3x3 kernel: everything else is an exact zero
    reach = np.zeros((H, W), bool)
    for dh in (-1, 0, 1):
        for dw in (-1, 0, 1):
            sh = np.zeros((H, W), bool)
            hs, he = max(0, -dh), min(H, H - dh)
            ws_, we = max(0, -dw), min(W, W - dw)
            sh[hs:he, ws_:we] = mask[hs + dh:he + dh, ws_ + dw:we + dw]
            reach |= sh
    assert np.all(got[:, ~reach, :] == 0.0)


def test_tr16_layout(ops):
    """ds_read_b64_tr_b16 over a ramp: lane l, element j must read lds[(l&15) + 16*j + 64*(l>>4)]."""
    out = ops.selftest_tr16()
    exp = np.array([[(l & 15) + 16 * j + 64 * (l >> 4) for j in range(4)] for l in range(64)])
    assert np.array_equal(out, exp), 'unexpected transpose-read layout:\n%s' % out


def _k_order(x, taps, ld):
    """[rows][taps][ld] (tap-major) -> the kernels' K order (include/yolo2_hip.h): 64-channel chunk, tap, channel in chunk
    when ld % 64 == 0 and taps > 1; otherwise unchanged."""
    rows = x.shape[0]
    if taps == 1 or ld % 64:
        return x.reshape(rows, taps * ld)
    return x.reshape(rows, taps, ld // 64, 64).transpose(0, 2, 1, 3).reshape(rows, taps * ld)


@pytest.mark.parametrize('dims', [(3, 5, 11, 8, 16), (3, 100, 70, 128, 72), (3, 64, 192, 64, 192), (1, 128, 64, 128, 64)])
def test_filter_prep_exact(ops, dims):
    rng = np.random.RandomState(0)
    k, Cin, Cout, ldcin, ldcout = dims
    taps = k * k
    w = rng.randn(k, k, Cin, Cout).astype(np.float32)
    Ff = torch.full((Cout * taps * ldcin,), 7.0, dtype=torch.float32, device='cuda')
    Fd = torch.full((Cin * taps * ldcout,), 7.0, dtype=torch.float32, device='cuda')
    ops.filter_prep(dev(w), Ff, Fd, k, Cin, ldcin, Cout, ldcout, torch.float32)
    torch.cuda.synchronize()
    ef = np.zeros((Cout, taps, ldcin), np.float32)
    ef[:, :, :Cin] = w.reshape(taps, Cin, Cout).transpose(2, 0, 1)
    ed = np.zeros((Cin, taps, ldcout), np.float32)
    ed[:, :, :Cout] = w[::-1, ::-1].reshape(taps, Cin, Cout).transpose(1, 0, 2)
    assert np.array_equal(host(Ff).reshape(Cout, -1), _k_order(ef, taps, ldcin))
    assert np.array_equal(host(Fd).reshape(Cin, -1), _k_order(ed, taps, ldcout))


def test_filter_prep_batch_matches_per_layer(ops):
    import ctypes
    from yolo_tf_amd._lib import FilterDesc
    rng = np.random.RandomState(1)
    layers = [(3, 3, 32), (3, 32, 64), (1, 128, 64), (3, 40, 125), (1, 1024, 125), (3, 128, 192)]
    for tdtype in (torch.float32, torch.bfloat16):
        arr = (FilterDesc * len(layers))()
        keep, first = [], 0
        for d, (k, cin, cout) in zip(arr, layers):
            ldcin, ldcout = ops.pad8(cin), ops.pad8(cout)
            w = dev(rng.randn(k, k, cin, cout).astype(np.float32))
            ff = torch.full((cout * k * k * ldcin,), 3.0, dtype=tdtype, device='cuda')
            fd = torch.full((cin * k * k * ldcout,), 3.0, dtype=tdtype, device='cuda')
            ef, ed = torch.zeros_like(ff), torch.zeros_like(fd)
            ops.filter_prep(w, ef, ed, k, cin, ldcin, cout, ldcout, tdtype)
            d.W, d.Ffwd, d.Fdgr = w.data_ptr(), ff.data_ptr(), fd.data_ptr()
            d.ksize, d.cin, d.ldcin, d.cout, d.ldcout, d.first_block = k, cin, ldcin, cout, ldcout, first
            first += ops.filter_prep_blocks(k, ldcin, ldcout)
            keep.append((w, ff, fd, ef, ed))
        descs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
        ops.filter_prep_batch(descs, len(layers), first, tdtype)
        torch.cuda.synchronize()
        for w, ff, fd, ef, ed in keep:
            assert torch.equal(ff, ef) and torch.equal(fd, ed)


def test_adam_fused_with_filter_prep_is_bit_identical(ops):
    """yolo2_adam_filter_prep == yolo2_adam over the arena followed by yolo2_filter_prep_batch, bit for bit (parameters, both Adam slots,
    both operand layouts), including the non-filter parameter ranges and the 16-byte alignment gaps (never touched); and the Adam
    arithmetic itself against the oracle."""
    from yolo_tf_amd._lib import FilterDesc
    rng = np.random.RandomState(4)
    layers = [(3, 3, 32), (3, 32, 64), (1, 128, 64), (3, 40, 125), (1, 1024, 125), (3, 128, 192)]
    # arena: per layer [filter][gamma-like range of cout][3 floats of padding]
    offs, off = [], 0
    for k, cin, cout in layers:
        offs.append((off, k * k * cin * cout, off + (k * k * cin * cout + 3) // 4 * 4, cout))
        off = offs[-1][2] + (cout + 3) // 4 * 4 + 4
    n = off
    hyper = dict(alpha=1e-3 * np.sqrt(1 - 0.999 ** 3) / (1 - 0.9 ** 3), b1=0.9, b2=0.999, eps=1e-8, gscale=0.5)
    for tdtype in (torch.float32, torch.bfloat16):
        p0 = rng.randn(n).astype(np.float32)
        g0, m0, v0 = rng.randn(n).astype(np.float32), rng.randn(n).astype(np.float32) * 0.1, (rng.rand(n) * 0.01).astype(np.float32)
        res = {}
        for fused in (False, True):
            P, G, M_, V = dev(p0), dev(g0), dev(m0), dev(v0)
            arr = (FilterDesc * len(layers))()
            keep, first, small = [], 0, []
            for d, (k, cin, cout), (o, sz, o2, n2) in zip(arr, layers, offs):
                ldcin, ldcout = ops.pad8(cin), ops.pad8(cout)
                ff = torch.full((cout * k * k * ldcin,), 3.0, dtype=tdtype, device='cuda')
                fd = torch.full((cin * k * k * ldcout,), 3.0, dtype=tdtype, device='cuda')
                d.W, d.Ffwd, d.Fdgr = P[o:].data_ptr(), ff.data_ptr(), fd.data_ptr()
                d.ksize, d.cin, d.ldcin, d.cout, d.ldcout, d.first_block = k, cin, ldcin, cout, ldcout, first
                first += ops.filter_prep_blocks(k, ldcin, ldcout)
                keep.append((ff, fd))
                small += [o2, n2]
            descs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
            if fused:
                ops.adam_filter_prep(descs, len(layers), first, torch.tensor(small, dtype=torch.int64, device='cuda'), len(layers), P, G, M_, V,
                                     hyper['alpha'], hyper['b1'], hyper['b2'], hyper['eps'], hyper['gscale'], tdtype)
            else:
                for o, sz, o2, n2 in offs:         # the plain kernel over exactly the same elements
                    ops.adam(P[o:], G[o:], M_[o:], V[o:], sz, hyper['alpha'], hyper['b1'], hyper['b2'], hyper['eps'], hyper['gscale'])
                    ops.adam(P[o2:], G[o2:], M_[o2:], V[o2:], n2, hyper['alpha'], hyper['b1'], hyper['b2'], hyper['eps'], hyper['gscale'])
                ops.filter_prep_batch(descs, len(layers), first, tdtype)
            torch.cuda.synchronize()
            res[fused] = (P, M_, V, keep)
        for what, a, b in zip(('params', 'm', 'v'), res[False][:3], res[True][:3]):
            assert torch.equal(a, b), (what, float((a - b).abs().max()), int((a != b).sum()))
        for (ff0, fd0), (ff1, fd1) in zip(res[False][3], res[True][3]):
            assert torch.equal(ff0, ff1) and torch.equal(fd0, fd1)
        touched = np.zeros(n, bool)
        for o, sz, o2, n2 in offs:
            touched[o:o + sz] = True
            touched[o2:o2 + n2] = True
        assert np.array_equal(host(res[True][0])[~touched], p0[~touched])           # alignment gaps untouched
        w_ref, m_ref, v_ref = R.adam_step(p0[touched], g0[touched] * np.float32(hyper['gscale']), m0[touched], v0[touched], 1e-3, 3)
        assert np.allclose(host(res[True][0])[touched], w_ref, rtol=2e-6, atol=1e-7)
        assert np.allclose(host(res[True][1])[touched], m_ref, rtol=2e-6, atol=1e-8) and np.allclose(host(res[True][2])[touched], v_ref, rtol=2e-6, atol=1e-10)


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(2, 26, 26, 32), (1, 52, 48, 64), (3, 8, 6, 8), (2, 14, 14, 512), (1, 4, 4, 1024)])
def test_bn_leaky_pool_fused_equals_unfused(ops, shape, mode):
    """yolo2_bn_leaky_pool / _bwd_reduce / _bwd_apply against the unfused chain bn_leaky -> maxpool_fwd and
    maxpool_bwd -> bn_leaky_bwd_reduce/apply (themselves checked against the oracle above), ties included."""
    B, H, W, C = shape
    tdtype = torch.float32 if mode == 'f32' else torch.bfloat16
    rng = np.random.RandomState(C + H)
    y = (rng.randn(B, H, W, C) * 1.5 + rng.randn(C)).astype(np.float32)
    y[:, ::2, ::2, : C // 2] = y[:, 1::2, ::2, : C // 2]        # exact ties inside many windows: first-max routing must agree
    if mode == 'bf16':
        y = bf16_round(y)
    gamma = (rng.rand(C) + 0.5).astype(np.float32) * np.where(rng.rand(C) < 0.2, -1, 1).astype(np.float32)   # some negative scales
    beta = (rng.randn(C) * 0.2).astype(np.float32)
    M, MP = B * H * W, B * (H // 2) * (W // 2)
    yd = dev(y, tdtype)
    mean, var = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    ws = torch.zeros(1026 * C + 64, dtype=torch.float64, device='cuda')
    ops.bn_stats(yd, mean, var, ws, M, C)
    g, b_ = dev(gamma), dev(beta)
    a = torch.zeros(M * C, dtype=tdtype, device='cuda')
    p_ref = torch.zeros(MP * C, dtype=tdtype, device='cuda')
    ops.bn_leaky(yd, mean, var, g, b_, a, M, C, C, 1e-5, 0.1)
    ops.maxpool_fwd(a, p_ref, B, H, W, C, 2)
    p = torch.zeros(MP * C, dtype=tdtype, device='cuda')
    idx = torch.full((MP * C,), 9, dtype=torch.uint8, device='cuda')
    ops.bn_leaky_pool(yd, mean, var, g, b_, p, idx, B, H, W, C, C, 1e-5, 0.1)
    torch.cuda.synchronize()
    assert torch.equal(p, p_ref)
    assert int(idx.max()) <= 3
    dp = rng.randn(B, H // 2, W // 2, C).astype(np.float32)
    if mode == 'bf16':
        dp = bf16_round(dp)
    dpd = dev(dp, tdtype)
    da = torch.zeros(M * C, dtype=tdtype, device='cuda')
    ops.maxpool_bwd(a, dpd, da, B, H, W, C, 2)
    dg_ref, db_ref = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    ops.bn_leaky_bwd_reduce(da, C, yd, mean, var, g, b_, dg_ref, db_ref, ws, M, C, 1e-5, 0.1)
    dy_ref = torch.zeros(M * C, dtype=tdtype, device='cuda')
    ops.bn_leaky_bwd_apply(da, C, yd, mean, var, g, b_, dg_ref, db_ref, dy_ref, M, C, 1e-5, 0.1)
    dg, db = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    ops.bn_leaky_pool_bwd_reduce(dpd, C, idx, yd, mean, var, g, b_, dg, db, ws, B, H, W, C, 1e-5, 0.1)
    dy = torch.full((M * C,), 7.0, dtype=tdtype, device='cuda')
    ops.bn_leaky_pool_bwd_apply(dpd, C, idx, yd, mean, var, g, b_, dg_ref, db_ref, dy, B, H, W, C, 1e-5, 0.1)
    torch.cuda.synchronize()
    assert_close(host(dg), host(dg_ref), 2e-5, 'fused dgamma %s %s' % (shape, mode))
    assert_close(host(db), host(db_ref), 2e-5, 'fused dbeta %s %s' % (shape, mode))
    assert_close(host(dy), host(dy_ref), 1e-6 if mode == 'f32' else 8e-3, 'fused dy %s %s' % (shape, mode))


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(2, 26, 26, 32), (4, 13, 13, 1024), (1, 52, 52, 64), (3, 7, 5, 8)])
def test_bn_leaky_forward_backward(ops, shape, mode):
    B, H, W, C = shape
    tdtype = torch.float32 if mode == 'f32' else torch.bfloat16
    rtol = F32_RTOL if mode == 'f32' else BF16_RTOL
    rng = np.random.RandomState(C)
    y = (rng.randn(B, H, W, C) * 1.5 + rng.randn(C)).astype(np.float32)
    da = rng.randn(B, H, W, C).astype(np.float32)
    gamma = (rng.rand(C) + 0.5).astype(np.float32)
    beta = (rng.randn(C) * 0.2).astype(np.float32)
    if mode == 'bf16':
        y, da = bf16_round(y), bf16_round(da)
    M = B * H * W
    mean_r, var_r = R.bn_moments(y)
    z = R.bn_apply(y, mean_r, var_r, gamma, beta)
    a_r = R.leaky_relu(z)
    dy_r, dg_r, db_r = R.bn_train_bwd(y, mean_r, var_r, gamma, R.leaky_relu_grad(z, da))

    yd, dad = dev(y, tdtype), dev(da, tdtype)
    mean, var = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    ws = torch.zeros(1026 * C, dtype=torch.float64, device='cuda')
    ops.bn_stats(yd, mean, var, ws, M, C)
    lda = C + 16
    A = torch.zeros(M * lda, dtype=tdtype, device='cuda')
    ops.bn_leaky(yd, mean, var, dev(gamma), dev(beta), A, M, C, lda, 1e-5, 0.1)
    dg, db = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    ops.bn_leaky_bwd_reduce(dad, C, yd, mean, var, dev(gamma), dev(beta), dg, db, ws, M, C, 1e-5, 0.1)
    dY = torch.zeros(M * C, dtype=tdtype, device='cuda')
    ops.bn_leaky_bwd_apply(dad, C, yd, mean, var, dev(gamma), dev(beta), dg, db, dY, M, C, 1e-5, 0.1)
    mm, mv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    ops.bn_ema(mm, mv, mean, var, C, 0.999)
    mm2, mv2, mean2, var2 = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda'), torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    ops.bn_stats_ema(yd, mean2, var2, mm2, mv2, 0.999, ws, M, C)          # fused statistics + moving-average update
    torch.cuda.synchronize()
    assert torch.equal(mean2, mean) and torch.equal(var2, var) and torch.equal(mm2, mm) and torch.equal(mv2, mv)
    torch.cuda.synchronize()
    assert_close(host(mean), mean_r, 1e-5, 'mean')
    assert_close(host(var), var_r, 1e-5, 'var')
    a = host(A).reshape(M, lda)
    assert np.all(a[:, C:] == 0)
    assert_close(a[:, :C].reshape(shape), a_r, rtol, 'bn_leaky')
    assert_close(host(dg), dg_r, max(rtol, 2e-4), 'dgamma')
    assert_close(host(db), db_r, max(rtol, 2e-4), 'dbeta')
    assert_close(host(dY).reshape(shape), dy_r, max(rtol, 2e-4), 'dY')
    assert_close(host(mm), R.bn_ema(np.zeros(C, np.float32), mean_r), 1e-5, 'ema mean')
    assert_close(host(mv), R.bn_ema(np.ones(C, np.float32), var_r), 1e-5, 'ema var')


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(2, 64, 96), (1, 32, 32), (3, 18, 40), (2, 12, 72), (2, 416, 416)])
def test_first_layer_wgrad_with_bn_backward_inside(ops, shape, mode):
    """yolo2_first_layer_wgrad_bn (conv_first_wgrad_bn_kernel: the image layer's output gradient formed in LDS from the raw forward output, the pooled
    gradient and the arg-max codes, and consumed there by the filter-gradient MFMAs) against the two launches it replaces --
    yolo2_bn_leaky_pool_bwd_apply_fin + yolo2_conv2d_wgrad, both pinned to the oracle by the tests above -- and, on the small shapes, the oracle's own
    bn_train_bwd / max_pool_grad / conv2d_wgrad chain.  Widths that are not a multiple of the 32-pixel segment (40, 72) exercise the dead pixels of a
    segment, which must contribute exactly nothing."""
    B, H, W = shape
    tdtype = torch.float32 if mode == 'f32' else torch.bfloat16
    q = (lambda a: a) if mode == 'f32' else bf16_round
    rng = np.random.RandomState(H * 7 + W)
    img = q(rng.randn(B, H, W, 3).astype(np.float32))
    x = dev(pad_channels(img, 8), tdtype)
    M, MP = B * H * W, B * (H // 2) * (W // 2)
    y_np = q((rng.randn(B, H, W, 32) * 1.5 + 0.3).astype(np.float32))
    y = dev(y_np, tdtype)
    mean, var = dev((rng.randn(32) * 0.3).astype(np.float32)), dev((rng.rand(32) + 0.5).astype(np.float32))
    gamma, beta = dev((rng.rand(32) + 0.5).astype(np.float32)), dev((rng.randn(32) * 0.2).astype(np.float32))
    # pooled activation + arg-max codes from the product's own forward (so that the codes are the ones a step would hold)
    P0 = torch.zeros(MP * 32, dtype=tdtype, device='cuda')
    idx = torch.full((MP * 32,), 9, dtype=torch.uint8, device='cuda')
    ops.bn_leaky_pool(y, mean, var, gamma, beta, P0, idx, B, H, W, 32, 32, 1e-5, 0.1)
    dp_np = q(rng.randn(B, H // 2, W // 2, 32).astype(np.float32))
    dpd = dev(dp_np, tdtype)
    ws = torch.zeros(ops.workspace_bytes('bn', 32) // 4 + 2 * 1024 * 32, dtype=torch.float32, device='cuda')
    limit = ops.bn_fin_rows_limit(32, tdtype)
    assert limit >= 128
    rows = ops.bn_leaky_pool_bwd_reduce_part(dpd, 32, idx, y, mean, var, gamma, beta, ws, limit, B, H, W, 32, 1e-5, 0.1)
    # ---- the two launches
    dg0, db0 = torch.zeros(32, device='cuda'), torch.zeros(32, device='cuda')
    dy0 = torch.zeros(M * 32, dtype=tdtype, device='cuda')
    ops.bn_leaky_pool_bwd_apply_fin(dpd, 32, idx, y, mean, var, gamma, beta, ws, rows, rows * 32, dg0, db0, dy0, B, H, W, 32, 1e-5, 0.1)
    dW0 = torch.zeros(9 * 3 * 32, dtype=torch.float32, device='cuda')
    ops.conv2d_wgrad(x, dy0, dW0, B, H, W, 3, 8, 32, 32, 3)
    # ---- one launch (+ a buffer to clear on the side, like the engine hands it)
    dg1, db1 = torch.zeros(32, device='cuda'), torch.zeros(32, device='cuda')
    dW1 = torch.zeros(9 * 3 * 32, dtype=torch.float32, device='cuda')
    junk = torch.full((4096,), 3.0, dtype=torch.float32, device='cuda')
    ops.first_layer_wgrad_bn(x, y, dpd, 32, idx, mean, var, gamma, beta, ws, rows, rows * 32, dg1, db1, dW1, B, H, W, 3, 1e-5, 0.1, junk, 4096)
    torch.cuda.synchronize()
    assert float(junk.abs().max()) == 0.0
    assert_close(host(dg1), host(dg0), 1e-6, 'dgamma')
    assert_close(host(db1), host(db0), 1e-6, 'dbeta')
    ref = host(dW0).reshape(3, 3, 3, 32)
    got = host(dW1).reshape(3, 3, 3, 32)
    # same operands up to the last bit of a few dY elements (the two compilations may contract the f32 multiply-adds differently), f32 sums in another order
    tol = 2e-5 if mode == 'f32' else 2e-3
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max(), 'dW fused vs two launches %s %s: %.3e of %.3e' % (shape, mode, np.abs(got - ref).max(), np.abs(ref).max())
    if M <= 20000:        # the oracle directly
        dgo, dbo = host(dg0).astype(np.float64), host(db0).astype(np.float64)
        mu, inv = host(mean).astype(np.float64), 1.0 / np.sqrt(host(var).astype(np.float64) + 1e-5)
        ga, bt = host(gamma).astype(np.float64), host(beta).astype(np.float64)
        y64 = y_np.astype(np.float64)
        xh = (y64 - mu) * inv
        z = xh * ga + bt
        # routed by the arg-max codes the forward stored (k = 2 * row parity + column parity; pinned to the oracle's first-maximum rule by the pool
        # tests above -- recomputing the arg-max here in f64 would differ from the bf16 forward wherever two rounded activations tie)
        codes = host(idx).reshape(B, H // 2, W // 2, 32)
        da = np.zeros((B, H, W, 32))
        for k in range(4):
            da[:, (k >> 1)::2, (k & 1)::2, :] = np.where(codes == k, dp_np.astype(np.float64), 0.0)
        g = np.where(z >= 0, da, 0.1 * da)
        dy = q(((ga * inv) * (g - dbo / M - xh * (dgo / M))).astype(np.float32))
        dWo = R.conv2d_wgrad(img.astype(np.float64), dy.astype(np.float64), 3, 3)
        assert np.abs(got - dWo).max() <= (2e-4 if mode == 'f32' else 4e-3) * np.abs(dWo).max(), 'dW fused vs oracle %s %s' % (shape, mode)


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(2, 64, 96), (1, 32, 32), (3, 18, 40), (2, 416, 416)])
def test_first_layer_fused_with_bn_leaky_pool(ops, shape, mode):
    """yolo2_first_layer_* (the image layer's output recomputed inside its consumers, never stored) against the stored-output path --
    yolo2_conv2d_bn + bn_finalize, bn_leaky_pool, bn_leaky_pool_bwd_reduce / _bwd_apply -- which the oracle tests above pin: pooled
    activation and arg-max bit for bit, statistics / dgamma / dbeta up to f32 summation order, dY to the last bit or two;
    plus the oracle directly on the small shapes."""
    B, H, W = shape
    tdtype = torch.float32 if mode == 'f32' else torch.bfloat16
    rng = np.random.RandomState(H + W)
    img = rng.randn(B, H, W, 3).astype(np.float32)
    w = (rng.randn(3, 3, 3, 32) * 0.3).astype(np.float32)
    if mode == 'bf16':
        img = bf16_round(img)
    x = dev(pad_channels(img, 8), tdtype)
    F = torch.zeros(32 * 9 * 8, dtype=tdtype, device='cuda')
    ops.filter_prep(dev(w), F, None, 3, 3, 8, 32, 32, tdtype)
    gamma, beta = dev((rng.rand(32) + 0.5).astype(np.float32)), dev((rng.randn(32) * 0.2).astype(np.float32))
    shift = dev((rng.randn(32) * 0.1).astype(np.float32))
    M, MP = B * H * W, B * (H // 2) * (W // 2)
    ws = torch.zeros(1024 + 256 * 256 * 128, dtype=torch.float32, device='cuda')
    red = torch.zeros(ops.workspace_bytes('bn', 32) // 8, dtype=torch.float64, device='cuda')
    # ---- stored-output path
    y = torch.zeros(M * 32, dtype=tdtype, device='cuda')
    part = torch.zeros(2 * 256 * 32, dtype=torch.float32, device='cuda')
    mean0, var0 = torch.zeros(32, device='cuda'), torch.zeros(32, device='cuda')
    ops.conv2d_bn(x, F, y, ws, B, H, W, 8, 8, 32, 32, 3, shift, part)
    assert ops.last_conv_plan()['BM'] == -1          # the direct image-layer kernel
    ops.bn_finalize(part, shift, M, 32, mean0, var0, None, None, 0.999)
    P0 = torch.zeros(MP * 32, dtype=tdtype, device='cuda')
    idx0 = torch.full((MP * 32,), 9, dtype=torch.uint8, device='cuda')
    ops.bn_leaky_pool(y, mean0, var0, gamma, beta, P0, idx0, B, H, W, 32, 32, 1e-5, 0.1)
    dp = rng.randn(B, H // 2, W // 2, 32).astype(np.float32)
    dpd = dev(bf16_round(dp) if mode == 'bf16' else dp, tdtype)
    dg0, db0 = torch.zeros(32, device='cuda'), torch.zeros(32, device='cuda')
    ops.bn_leaky_pool_bwd_reduce(dpd, 32, idx0, y, mean0, var0, gamma, beta, dg0, db0, red, B, H, W, 32, 1e-5, 0.1)
    dy0 = torch.zeros(M * 32, dtype=tdtype, device='cuda')
    ops.bn_leaky_pool_bwd_apply(dpd, 32, idx0, y, mean0, var0, gamma, beta, dg0, db0, dy0, B, H, W, 32, 1e-5, 0.1)
    # ---- fused path
    mean1, var1 = torch.zeros(32, device='cuda'), torch.zeros(32, device='cuda')
    ops.first_layer_stats(x, F, B, H, W, shift, part)
    ops.bn_finalize(part, shift, M, 32, mean1, var1, None, None, 0.999)
    P1 = torch.zeros(MP * 32, dtype=tdtype, device='cuda')
    idx1 = torch.full((MP * 32,), 9, dtype=torch.uint8, device='cuda')
    ops.first_layer_bn_leaky_pool(x, F, mean0, var0, gamma, beta, P1, idx1, B, H, W, 32, 1e-5, 0.1)       # (same moments as the stored path: bit-exact comparison)
    dg1, db1 = torch.zeros(32, device='cuda'), torch.zeros(32, device='cuda')
    ops.first_layer_pool_bwd_reduce(x, F, dpd, 32, idx0, mean0, var0, gamma, beta, part, B, H, W, 1e-5, 0.1)
    ops.bn_part_to_grads(part, 32, dg1, db1)
    dy1 = torch.full((M * 32,), 7.0, dtype=tdtype, device='cuda')
    ops.first_layer_pool_bwd_apply(x, F, dpd, 32, idx0, mean0, var0, gamma, beta, dg0, db0, dy1, B, H, W, 1e-5, 0.1)
    P2 = torch.zeros(MP * 32, dtype=tdtype, device='cuda')
    ops.first_layer_bn_leaky_pool(x, F, mean0, var0, gamma, beta, P2, None, B, H, W, 32, 1e-5, 0.1)         # inference form: no arg-max output
    torch.cuda.synchronize()
    assert float(part.abs().max()) == 0.0
    assert_close(host(mean1), host(mean0), 2e-6, 'mean')
    assert_close(host(var1), host(var0), 2e-5, 'var')
    assert torch.equal(P1, P0) and torch.equal(idx1, idx0) and torch.equal(P2, P0)
    assert_close(host(dg1), host(dg0), 2e-5, 'dgamma')
    assert_close(host(db1), host(db0), 2e-5, 'dbeta')
    # (dY goes through a few f32 multiply-adds that the two kernels' compilations may contract differently: last-bit agreement)
    assert_close(host(dy1), host(dy0), 2e-6 if mode == 'f32' else 8e-3, 'dY')
    assert float((dy1.float() - dy0.float()).abs().mean()) <= (1e-7 if mode == 'f32' else 2e-4) * float(dy0.float().abs().mean() + 1e-30)
    if M <= 20000:        # the oracle directly
        yq = R.conv2d(img, bf16_round(w) if mode == 'bf16' else w)
        if mode == 'bf16':
            yq = bf16_round(yq)
        a = R.leaky_relu(R.bn_apply(yq, host(mean0), host(var0), host(gamma), host(beta)))
        if mode == 'bf16':
            a = bf16_round(a)
        assert_close(host(P1).reshape(B, H // 2, W // 2, 32), R.max_pool(a, 2), F32_RTOL if mode == 'f32' else BF16_RTOL, 'pooled activation vs oracle')


def _host_partials(vals0, vals1, rows, C, plane_rows, rng, poison):
    """[2][plane_rows][C] f32 partial rows: the M samples of each channel dealt to ``rows`` groups at random (what a producer's tiles
    leave behind); rows >= ``rows`` hold NaN when ``poison`` (a consumer must not read them)."""
    M = vals0.shape[0]
    grp = rng.randint(0, rows, M)
    part = np.full((2, plane_rows, C), np.nan if poison else 0.0, np.float32)
    part[:, :rows] = 0
    np.add.at(part[0], grp, vals0.astype(np.float32))
    np.add.at(part[1], grp, vals1.astype(np.float32))
    return part


def test_bn_leaky_fin_shift_is_read_only_across_thousands_of_workgroups(ops):
    """Round-3 advisor finding: the engine handed the LIVE moving mean to yolo2_bn_leaky_fin both as `shift` (read by every workgroup
    after its prologue) and as `moving_mean` (updated in place by the first workgroup of each channel slice) -- late workgroups
    normalised with a mean off by (1 - decay) * (batch mean - moving mean).  The ABI now rejects the aliased call; with the snapshot
    the engine passes (engine.Engine.state_snap) a launch of ~2700 workgroups per slice equals yolo2_bn_finalize + yolo2_bn_leaky
    (to an f32 ulp of the moments) and is deterministic, with a decay small enough that a stale read would move the mean by 30 % of the offset."""
    B, H, W, C, rows = 16, 104, 104, 128, 16
    M = B * H * W
    T = torch.bfloat16
    rng = np.random.RandomState(11)
    y = bf16_round((rng.randn(M, C) * 1.5 + rng.randn(C) * 2.0).astype(np.float32))
    mm0 = (y.mean(0) + 3.0 + rng.randn(C)).astype(np.float32)       # a moving mean far from the batch mean: a stale read is visible
    d = y - mm0
    part_all = _host_partials(d, d * d, rows, C, 256, rng, poison=False)     # (yolo2_bn_finalize sums all 256 rows: the unused ones are zero)
    part = part_all.copy()
    part[:, rows:] = np.nan                                                  # ... the folded form must not read beyond `rows`
    yd, g, b_ = dev(y, T), dev((rng.rand(C) + 0.5).astype(np.float32)), dev((rng.randn(C) * 0.2).astype(np.float32))
    decay = 0.7
    # two-launch reference: finalize (zeroes its partial rows) + apply
    mean_r, var_r = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    mm_r, mv_r = dev(mm0), torch.ones(C, device='cuda')
    A_r = torch.zeros(M * C, dtype=T, device='cuda')
    ops.bn_finalize(dev(part_all), dev(mm0), M, C, mean_r, var_r, mm_r, mv_r, decay)
    ops.bn_leaky(yd, mean_r, var_r, g, b_, A_r, M, C, C, 1e-5, 0.1)
    for rep in range(3):
        mean, var = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
        mm, mv = dev(mm0), torch.ones(C, device='cuda')
        A = torch.zeros(M * C, dtype=T, device='cuda')
        ops.bn_leaky_fin(yd, dev(part), rows, dev(mm0), mean, var, mm, mv, decay, g, b_, A, M, C, C, 1e-5, 0.1)
        torch.cuda.synchronize()
        # (the two forms add the same f64 partial sums in different orders: equal to an f32 ulp, in practice identical)
        np.testing.assert_allclose(host(mean), host(mean_r), rtol=3e-7, atol=0)
        np.testing.assert_allclose(host(var), host(var_r), rtol=3e-7, atol=0)
        np.testing.assert_allclose(host(mm), host(mm_r), rtol=3e-7, atol=0)
        np.testing.assert_allclose(host(mv), host(mv_r), rtol=3e-7, atol=0)
        nbad = int((A != A_r).sum())
        assert nbad <= (0 if torch.equal(mean, mean_r) and torch.equal(var, var_r) else M * C // 1000), \
            'repeat %d: %d elements differ from the two-launch form' % (rep, nbad)
        if rep == 0:
            A0 = A
        assert torch.equal(A, A0), 'repeat %d differs from repeat 0: the launch is not deterministic' % rep
    mm = dev(mm0)
    with pytest.raises(RuntimeError):
        ops.bn_leaky_fin(yd, dev(part), rows, mm, mean, var, mm, mv, decay, g, b_, A, M, C, C, 1e-5, 0.1)
    P = torch.zeros(B * (H // 2) * (W // 2) * C, dtype=T, device='cuda')
    with pytest.raises(RuntimeError):
        ops.bn_leaky_pool_fin(yd, dev(part), rows, mm, mean, var, mm, mv, decay, g, b_, P, None, B, H, W, C, C, 1e-5, 0.1)


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
@pytest.mark.parametrize('shape,rows', [((2, 26, 26, 32), 256), ((4, 13, 13, 1024), 44), ((1, 52, 52, 64), 128), ((2, 26, 26, 512), 16), ((3, 8, 6, 8), 5),
                                        ((2, 14, 14, 256), 64)])
def test_bn_consumers_with_folded_finalisation_vs_oracle(ops, shape, rows, mode):
    """yolo2_bn_leaky_fin / _pool_fin / _bwd_apply_fin / _pool_bwd_apply_fin: the kernels that sum the partial rows in their own prologue,
    against the ORACLE (moments, moving averages, BN + leaky, pool, dgamma / dbeta, dY), plus the side clearing of another buffer."""
    B, H, W, C = shape
    tdtype = torch.float32 if mode == 'f32' else torch.bfloat16
    rtol = F32_RTOL if mode == 'f32' else BF16_RTOL
    rng = np.random.RandomState(C + rows)
    y = (rng.randn(B, H, W, C) * 1.5 + rng.randn(C)).astype(np.float32)
    da = rng.randn(B, H, W, C).astype(np.float32)
    gamma = (rng.rand(C) + 0.5).astype(np.float32)
    beta = (rng.randn(C) * 0.2).astype(np.float32)
    shift = (rng.randn(C) * 0.3).astype(np.float32)
    if mode == 'bf16':
        y, da = bf16_round(y), bf16_round(da)
    M = B * H * W
    assert ops.bn_fin_supported(rows, C, tdtype)
    mean_r, var_r = R.bn_moments(y)
    z = R.bn_apply(y, mean_r, var_r, gamma, beta)
    a_r = R.leaky_relu(z)
    d = y.reshape(M, C) - shift
    part = _host_partials(d, d * d, rows, C, 256, rng, poison=True)
    yd, g, b_ = dev(y, tdtype), dev(gamma), dev(beta)
    mean, var = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    mm, mv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    lda = C + 16
    A = torch.zeros(M * lda, dtype=tdtype, device='cuda')
    other = torch.ones(4096, device='cuda')
    ops.bn_leaky_fin(yd, dev(part), rows, dev(shift), mean, var, mm, mv, 0.999, g, b_, A, M, C, lda, 1e-5, 0.1, other, 1024)
    torch.cuda.synchronize()
    assert_close(host(mean), mean_r, 1e-5, 'mean')
    assert_close(host(var), var_r, 2e-5, 'var')
    assert_close(host(mm), R.bn_ema(np.zeros(C, np.float32), mean_r), 1e-5, 'ema mean')
    assert_close(host(mv), R.bn_ema(np.ones(C, np.float32), var_r), 2e-5, 'ema var')
    a = host(A).reshape(M, lda)
    assert np.all(a[:, C:] == 0)
    assert_close(a[:, :C].reshape(shape), a_r, rtol, 'bn_leaky_fin')
    o = host(other)
    assert np.all(o[:1024] == 0) and np.all(o[1024:] == 1)
    # backward: partial rows of sum(g * xhat), sum(g) with g = dA * leaky'(z), in a compact [2][rows][C] workspace
    dz = R.leaky_relu_grad(z, da)
    dy_r, dg_r, db_r = R.bn_train_bwd(y, mean_r, var_r, gamma, dz)
    inv = 1.0 / np.sqrt(var_r + np.float32(1e-5))
    xhat = ((y - mean_r) * inv).reshape(M, C)
    bpart = _host_partials(dz.reshape(M, C) * xhat, dz.reshape(M, C), rows, C, rows, rng, poison=False)
    dg, db = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    dY = torch.zeros(M * C, dtype=tdtype, device='cuda')
    ops.bn_leaky_bwd_apply_fin(dev(da, tdtype), C, yd, dev(mean_r), dev(var_r), g, b_, dev(bpart), rows, rows * C, dg, db, dY, M, C, 1e-5, 0.1)
    torch.cuda.synchronize()
    assert_close(host(dg), dg_r, max(rtol, 2e-4), 'dgamma')
    assert_close(host(db), db_r, max(rtol, 2e-4), 'dbeta')
    assert_close(host(dY).reshape(shape), dy_r, max(rtol, 2e-4), 'dY (bwd_apply_fin)')
    limit = ops.bn_fin_rows_limit(C, tdtype)
    ws = torch.zeros(1026 * C + 64, dtype=torch.float64, device='cuda')
    r3 = ops.bn_leaky_bwd_reduce_part(dev(da, tdtype), C, yd, mean, var, g, b_, ws, limit, M, C, 1e-5, 0.1)
    dg3, db3 = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    dy3 = torch.zeros(M * C, dtype=tdtype, device='cuda')
    ops.bn_leaky_bwd_apply_fin(dev(da, tdtype), C, yd, mean, var, g, b_, ws, r3, r3 * C, dg3, db3, dy3, M, C, 1e-5, 0.1)
    torch.cuda.synchronize()
    assert_close(host(dg3), dg_r, max(rtol, 2e-4), 'dgamma (reduce_part)')
    assert_close(host(dy3).reshape(shape), dy_r, max(rtol, 2e-4), 'dY (reduce_part)')
    if H % 2 or W % 2:
        return
    # pooled forward: equals the pool of the rounded activation the plain form stores (first-max routing checked against yolo2_maxpool_fwd elsewhere)
    MP = B * (H // 2) * (W // 2)
    P = torch.zeros(MP * C, dtype=tdtype, device='cuda')
    idx = torch.full((MP * C,), 9, dtype=torch.uint8, device='cuda')
    mean2, var2 = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    ymax = torch.full((MP * C,), 5.0, dtype=tdtype, device='cuda')
    ops.bn_leaky_pool_fin(yd, dev(part), rows, dev(shift), mean2, var2, None, None, 0.999, g, b_, P, idx, B, H, W, C, C, 1e-5, 0.1, ymax=ymax)
    torch.cuda.synchronize()
    assert torch.equal(mean2, mean) and torch.equal(var2, var)
    # ymax = the stored convolution output at the recorded arg-max position, exactly
    yw = host(yd).reshape(B, H // 2, 2, W // 2, 2, C).transpose(0, 1, 3, 2, 4, 5).reshape(B, H // 2, W // 2, 4, C)
    pick = np.take_along_axis(yw, host(idx).reshape(B, H // 2, W // 2, 1, C).astype(np.int64), axis=3)[:, :, :, 0, :]
    assert np.array_equal(host(ymax).reshape(pick.shape), pick), 'ymax is not y at the arg-max'
    a_st = bf16_round(a_r) if mode == 'bf16' else a_r
    p_r = R.max_pool(a_st, 2)
    assert_close(host(P).reshape(p_r.shape), p_r, rtol, 'bn_leaky_pool_fin')
    assert int(idx.max()) <= 3
    # pooled backward through the device reduction (reduce_part leaves [2][rows][C]) == the plain pooled pair
    dp = rng.randn(B, H // 2, W // 2, C).astype(np.float32)
    if mode == 'bf16':
        dp = bf16_round(dp)
    dpd = dev(dp, tdtype)
    dg_ref, db_ref = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    ops.bn_leaky_pool_bwd_reduce(dpd, C, idx, yd, mean, var, g, b_, dg_ref, db_ref, ws, B, H, W, C, 1e-5, 0.1)
    dy_ref = torch.zeros(M * C, dtype=tdtype, device='cuda')
    ops.bn_leaky_pool_bwd_apply(dpd, C, idx, yd, mean, var, g, b_, dg_ref, db_ref, dy_ref, B, H, W, C, 1e-5, 0.1)
    r2 = ops.bn_leaky_pool_bwd_reduce_part(dpd, C, idx, yd, mean, var, g, b_, ws, limit, B, H, W, C, 1e-5, 0.1)
    assert 1 <= r2 <= limit
    dg2, db2 = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    dy2 = torch.full((M * C,), 7.0, dtype=tdtype, device='cuda')
    ops.bn_leaky_pool_bwd_apply_fin(dpd, C, idx, yd, mean, var, g, b_, ws, r2, r2 * C, dg2, db2, dy2, B, H, W, C, 1e-5, 0.1)
    torch.cuda.synchronize()
    assert_close(host(dg2), host(dg_ref), 2e-5, 'pooled dgamma')
    assert_close(host(db2), host(db_ref), 2e-5, 'pooled dbeta')
    assert_close(host(dy2), host(dy_ref), 1e-5 if mode == 'f32' else 8e-3, 'pooled dY')
    # the same sums from (dP, ymax) at the pooled resolution: what the engine runs (neither Y nor idx are read by the reduction)
    r3 = ops.bn_leaky_bwd_reduce_part(dpd, C, ymax, mean, var, g, b_, ws, limit, MP, C, 1e-5, 0.1)
    dg4, db4 = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    dy4 = torch.full((M * C,), 7.0, dtype=tdtype, device='cuda')
    ops.bn_leaky_pool_bwd_apply_fin(dpd, C, idx, yd, mean, var, g, b_, ws, r3, r3 * C, dg4, db4, dy4, B, H, W, C, 1e-5, 0.1)
    torch.cuda.synchronize()
    assert_close(host(dg4), host(dg_ref), 2e-5, 'pooled dgamma from ymax')
    assert_close(host(db4), host(db_ref), 2e-5, 'pooled dbeta from ymax')
    assert_close(host(dy4), host(dy_ref), 1e-5 if mode == 'f32' else 8e-3, 'pooled dY from ymax')



@pytest.mark.parametrize('mode', ['f32', 'bf16'])
@pytest.mark.parametrize('stride,shape', [(2, (2, 8, 12, 16)), (2, (1, 26, 26, 512)), (1, (2, 13, 13, 64)), (1, (1, 5, 3, 8))])
def test_maxpool(ops, stride, shape, mode):
    B, H, W, C = shape
    tdtype = torch.float32 if mode == 'f32' else torch.bfloat16
    rng = np.random.RandomState(H * W)
    a = rng.randint(-3, 4, size=shape).astype(np.float32)   # many ties: exercises first-max routing
    a[0, 0, :, :] = rng.randn(W, C)
    if mode == 'bf16':
        a = bf16_round(a)
    OH, OW = (H // 2, W // 2) if stride == 2 else (H, W)
    dp = bf16_round(rng.randn(B, OH, OW, C))
    P = torch.zeros(B * OH * OW * C, dtype=tdtype, device='cuda')
    dA = torch.full((B * H * W * C,), 9.0, dtype=tdtype, device='cuda')
    ops.maxpool_fwd(dev(a, tdtype), P, B, H, W, C, stride)
    ops.maxpool_bwd(dev(a, tdtype), dev(dp, tdtype), dA, B, H, W, C, stride)
    torch.cuda.synchronize()
    assert np.array_equal(host(P).reshape(B, OH, OW, C), R.max_pool(a, stride))
    ref = R.max_pool_grad(a, dp, stride)
    if mode == 'bf16' and stride == 1:
        ref = bf16_round(ref)   # overlapping windows sum up to 4 bf16 terms, stored as bf16
        assert_close(host(dA).reshape(shape), ref, 1e-2, 'pool bwd')
    else:
        assert np.array_equal(host(dA).reshape(shape), ref)
    if stride == 2:
        # accumulating form (second writer of a gradient): bit-equal to the temporary + yolo2_add_inplace pair it replaces
        first = bf16_round(rng.randn(*shape))
        g_pair, g_acc = dev(first, tdtype), dev(first, tdtype)
        ops.add_inplace(g_pair, dA, dA.numel())
        ops.maxpool_bwd_acc(dev(a, tdtype), dev(dp, tdtype), g_acc, B, H, W, C)
        torch.cuda.synchronize()
        assert torch.equal(g_pair, g_acc)
        want = first + ref
        assert_close(host(g_acc).reshape(shape), bf16_round(want) if mode == 'bf16' else want, 1e-6 if mode == 'f32' else 1e-2, 'pool bwd acc')


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_reorg_bit_exact(ops, golden_dir, mode):
    tdtype = torch.float32 if mode == 'f32' else torch.bfloat16
    rng = np.random.RandomState(0)
    B, H, W, C = 2, 26, 26, 512
    x = bf16_round(rng.randn(B, H, W, C))
    ldo = 4 * C + 1024
    out = torch.zeros(B * (H // 2) * (W // 2) * ldo, dtype=tdtype, device='cuda')
    ops.reorg(dev(x, tdtype), out, B, H, W, C, ldo)
    back = torch.zeros(B * H * W * C, dtype=tdtype, device='cuda')
    ops.reorg_bwd(out, ldo, back, B, H, W, C)
    torch.cuda.synchronize()
    o = host(out).reshape(B, H // 2, W // 2, ldo)
    assert np.array_equal(o[..., :4 * C], R.reorg(x))
    assert np.all(o[..., 4 * C:] == 0)
    assert np.array_equal(host(back).reshape(x.shape), x)
    # the reference's own known-answer test (model/yolo2/function.py:32-47), channels widened to the vector width
    g = np.load(os.path.join(golden_dir, 'labels.npz'))
    img = np.repeat(g['reorg/kat_in'].astype(np.float32), 8, axis=3)
    o2 = torch.zeros(1 * 2 * 2 * 32, dtype=tdtype, device='cuda')
    ops.reorg(dev(img, tdtype), o2, 1, 4, 4, 8, 32)
    torch.cuda.synchronize()
    o2 = host(o2).reshape(2, 2, 4, 8)
    for i in range(4):
        assert np.unique(o2[:, :, i, :]).tolist() == [float(g['reorg/kat_channel_values'][i])]


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
@pytest.mark.parametrize('M,C,ld', [(2704, 125, 128), (1352, 425, 432), (8192, 30, 32), (8193, 30, 32), (40000, 125, 128), (5, 8, 8)])
def test_bias_grad_both_forms(ops, M, C, ld, mode):
    """yolo2_bias_grad: the single-launch form (<= 8192 rows: the detection heads of the bench) and the two-stage form, vs the oracle's column sum."""
    tdtype = torch.float32 if mode == 'f32' else torch.bfloat16
    rng = np.random.RandomState(M + C)
    dy = np.zeros((M, ld), np.float32)
    dy[:, :C] = rng.randn(M, C) * (1.0 + np.arange(C) % 7)
    if mode == 'bf16':
        dy = bf16_round(dy)
    db = torch.full((C + 3,), 7.0, device='cuda')
    ws = torch.zeros(1024 * ld, dtype=torch.float64, device='cuda')
    ops.bias_grad(dev(dy, tdtype), ld, db, ws, M, C)
    torch.cuda.synchronize()
    ref = dy[:, :C].astype(np.float64).sum(0)
    got = host(db)
    assert np.all(got[C:] == 7.0), 'wrote past C'
    assert float(np.abs(got[:C] - ref).max()) <= 5e-6 * float(np.abs(dy).sum(0).max()) + 1e-6


def test_copy_add_biasgrad(ops):
    rng = np.random.RandomState(1)
    M, C = 338, 125
    ld = 128
    dy = np.zeros((M, ld), np.float32)
    dy[:, :C] = rng.randn(M, C)
    db = torch.zeros(C, device='cuda')
    ws = torch.zeros(512 * ld, dtype=torch.float64, device='cuda')
    ops.bias_grad(dev(dy), ld, db, ws, M, C)
    dst = torch.zeros(M * 256, device='cuda')
    ops.copy_channels(dev(dy), ld, dst, 256, M, 64)
    a, b = rng.randn(4096).astype(np.float32), rng.randn(4096).astype(np.float32)
    ad = dev(a)
    ops.add_inplace(ad, dev(b), 4096)
    torch.cuda.synchronize()
    assert_close(host(db), dy[:, :C].sum(0), 1e-5, 'bias_grad')
    d = host(dst).reshape(M, 256)
    assert np.array_equal(d[:, :64], dy[:, :64]) and np.all(d[:, 64:] == 0)
    assert np.array_equal(host(ad), a + b)


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_image_prep(ops, golden_dir, mode):
    tdtype = torch.float32 if mode == 'f32' else torch.bfloat16
    g = np.load(os.path.join(golden_dir, 'labels.npz'))
    img = np.stack([g['std/in'], g['std/in'][::-1] * 0.5 + 3])
    B, H, W, _ = img.shape
    out = torch.zeros(B * H * W * 8, dtype=tdtype, device='cuda')
    ws = torch.full((ops.workspace_bytes('image_prep', B) // 8,), 1e30, dtype=torch.float64, device='cuda')      # any content: no zeroing contract
    ops.image_prep(dev(img), out, ws, B, H * W, 0)
    torch.cuda.synchronize()
    o = host(out).reshape(B, H, W, 8)
    assert np.all(o[..., 3:] == 0)
    ref = np.stack([R.per_image_standardization(i) for i in img])
    assert_close(o[..., :3], ref, F32_RTOL if mode == 'f32' else 1e-2, 'standardize')
    assert_close(o[0, ..., :3], g['std/out'], F32_RTOL if mode == 'f32' else 1e-2, 'standardize vs reference golden')
    ops.image_prep(dev(img), out, ws, B, H * W, 1)
    torch.cuda.synchronize()
    assert_close(host(out).reshape(B, H, W, 8)[..., :3], img / 255., F32_RTOL if mode == 'f32' else 1e-2, 'darknet preprocess')


def _labels(rng, B, classes, cw, ch):
    outs = []
    for _ in range(B):
        k = rng.randint(1, 7)
        cen = rng.uniform(0.05, 0.95, (k, 2))
        wh = rng.uniform(0.05, 0.6, (k, 2))
        coord = np.clip(np.concatenate([cen - wh / 2, cen + wh / 2], 1), 0, 1).astype(np.float32)
        outs.append(R.transform_labels(rng.randint(0, classes, k), coord, classes, cw, ch))
    return tuple(np.stack([o[i] for o in outs]) for i in range(6))


VOC_ANCHORS = np.array([[1.08, 1.19], [3.42, 4.41], [6.63, 11.38], [9.42, 5.11], [16.62, 10.52]], np.float32)


@pytest.mark.parametrize('B,classes', [(2, 20), (16, 20), (3, 80)])
def test_loss_forward_backward_f32(ops, B, classes):
    rng = np.random.RandomState(B + classes)
    ch = cw = 13
    A = 5
    D = A * (5 + classes)
    ld = ops.pad8(D)
    net = (rng.randn(B, ch, cw, D) * 0.7).astype(np.float32)
    labels = _labels(rng, B, classes, cw, ch)
    hp = {'iou_best': 5., 'iou_normal': 1., 'coords': 1., 'prob': 1.}
    m = R.model_decode(net, classes, VOC_ANCHORS, training=True)
    obj, aux = R.objectives(m, labels)
    dnet = R.loss_backward(m, labels, aux, hp, classes)

    logits = dev(pad_channels(net, ld))
    dl = torch.full((B * ch * cw * ld,), 3.0, device='cuda')
    objs = torch.zeros(4, device='cuda')
    ws = torch.zeros(ops.loss_ws_floats(B, ch * cw, A), device='cuda')
    lab = [dev(l.reshape(B, ch * cw, -1)) for l in labels]
    ops.loss(logits, ld, dev(VOC_ANCHORS), lab, [hp[k] for k in R.OBJECTIVE_KEYS], objs, dl, ws, B, ch, cw, A, classes)
    torch.cuda.synchronize()
    got = host(objs)
    for i, k in enumerate(R.OBJECTIVE_KEYS):
        assert abs(got[i] - obj[k]) <= F32_RTOL * abs(obj[k]) + 1e-9, (k, got[i], obj[k])
    d = host(dl).reshape(B, ch, cw, ld)
    assert np.all(d[..., D:] == 0)
    assert_close(d[..., :D], dnet, F32_RTOL, 'dlogits')


def test_loss_known_answer_zero_logits(ops):
    """zero logits, no objects -> iou_normal = 0.25, rest 0."""
    B, ch, cw, A, C = 2, 13, 13, 5, 20
    ld = 128
    z = lambda *s: torch.zeros(*s, device='cuda')
    objs = z(4)
    ops.loss(z(B * 169 * ld), ld, dev(VOC_ANCHORS), [z(B * 169), z(B * 169 * C), z(B * 169 * 4), z(B * 169 * 2), z(B * 169 * 2), z(B * 169)],
             [5., 1., 1., 1.], objs, None, z(ops.loss_ws_floats(B, 169, A)), B, ch, cw, A, C)
    torch.cuda.synchronize()
    got = host(objs)
    assert got[0] == 0 and got[2] == 0 and got[3] == 0 and abs(got[1] - 0.25) < 1e-6


@pytest.mark.parametrize('classes', [20, 80])
def test_head_decode_f32(ops, classes):
    rng = np.random.RandomState(classes)
    B, ch, cw, A = 3, 13, 13, 5
    D = A * (5 + classes)
    ld = ops.pad8(D)
    net = (rng.randn(B, ch, cw, D)).astype(np.float32)
    m = R.model_decode(net, classes, VOC_ANCHORS, training=False)
    conf = torch.zeros(B * 169 * A * classes, device='cuda')
    mn, mx = torch.zeros(B * 169 * A * 2, device='cuda'), torch.zeros(B * 169 * A * 2, device='cuda')
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    ops.head_decode(dev(pad_channels(net, ld)), ld, dev(VOC_ANCHORS), conf, mn, mx, flag, B, ch, cw, A, classes)
    torch.cuda.synchronize()
    assert_close(host(conf).reshape(m['conf'].shape), m['conf'], F32_RTOL, 'conf')
    assert_close(host(mn).reshape(m['xy_min'].shape), m['xy_min'], F32_RTOL, 'xy_min')
    assert_close(host(mx).reshape(m['xy_max'].shape), m['xy_max'], F32_RTOL, 'xy_max')
    assert int(flag.item()) == 0
    bad = net.copy()
    bad[0, 0, 0, 3] = np.inf
    ops.head_decode(dev(pad_channels(bad, ld)), ld, dev(VOC_ANCHORS), conf, mn, mx, flag, B, ch, cw, A, classes)
    torch.cuda.synchronize()
    assert int(flag.item()) != 0       # tf.check_numerics counterpart (detect.py:70)


NMS_CASES = ['sparse20', 'sparse80', 'dense', 'dense845', 'clustered', 'identical_ties', 'ties', 'at_threshold', 'all_below', 'zero_area']


def _gpu_nms(ops, conf, mn, mx, thr, thr_iou):
    B, N, C = conf.shape
    cd = dev(conf)
    order = torch.zeros(B * N, dtype=torch.int32, device='cuda')
    ws = torch.zeros(B * N * C, dtype=torch.int32, device='cuda')
    ops.nms(cd, dev(mn), dev(mx), order, ws, B, N, C, thr, thr_iou)
    torch.cuda.synchronize()
    return host(cd).reshape(B, N, C), order.cpu().numpy().reshape(B, N)


def test_nms_reference_goldens_bit_exact(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, 'nms.npz'))
    for case in NMS_CASES:
        conf = g[case + '/conf_in']
        cells, A, C = conf.shape
        got, order = _gpu_nms(ops, conf.reshape(1, cells * A, C), g[case + '/xy_min'].reshape(1, -1, 2), g[case + '/xy_max'].reshape(1, -1, 2),
                              float(g[case + '/thr']), float(g[case + '/thr_iou']))
        assert np.array_equal(got[0], g[case + '/conf_out'].reshape(cells * A, C)), case
        assert np.array_equal(order[0], g[case + '/order']), case


def _c_nms(conf, mn, mx, thr, thr_iou):
    lib = ctypes.CDLL(os.path.join(ROOT, 'oracle', 'libnms_ref.so'))
    n, c = conf.shape
    conf = np.ascontiguousarray(conf, np.float32).copy()
    order = np.zeros(n, np.int64)
    P = ctypes.POINTER(ctypes.c_float)
    lib.nms_ref(conf.ctypes.data_as(P), np.ascontiguousarray(mn, np.float32).ctypes.data_as(P), np.ascontiguousarray(mx, np.float32).ctypes.data_as(P),
                ctypes.c_long(n), ctypes.c_long(c), ctypes.c_float(thr), ctypes.c_float(thr_iou), order.ctypes.data_as(ctypes.POINTER(ctypes.c_long)))
    return conf, order


@pytest.mark.parametrize('B,N,C,dense', [(8, 845, 20, False), (4, 845, 80, False), (2, 845, 20, True), (2, 1805, 20, False), (3, 100, 3, True),
                                         (1, 4096, 4, False), (1, 4001, 2, True)])      # the documented N limit (129 KB of LDS)
def test_nms_batched_vs_c_oracle(ops, B, N, C, dense):
    rng = np.random.RandomState(N + C)
    if dense:
        conf = rng.uniform(0, 0.5, (B, N, C)).astype(np.float32)
        conf[rng.rand(B, N, C) < 0.1] = 0.4          # ties
    else:
        conf = rng.uniform(0, 0.05, (B, N, C)).astype(np.float32)
        for b in range(B):
            hot = rng.choice(N, 40, replace=False)
            conf[b, hot, rng.randint(0, C, 40)] = rng.uniform(0.3, 0.95, 40).astype(np.float32)
    cen = rng.uniform(0, 13, (B, N, 2)).astype(np.float32)
    wh = rng.uniform(0.5, 5.5, (B, N, 2)).astype(np.float32)
    mn, mx = (cen - wh / 2).astype(np.float32), (cen + wh / 2).astype(np.float32)
    got, order = _gpu_nms(ops, conf, mn, mx, 0.3, 0.4)
    for b in range(B):
        rc, ro = _c_nms(conf[b], mn[b], mx[b], 0.3, 0.4)
        assert np.array_equal(got[b], rc), 'image %d: %d scores differ' % (b, (got[b] != rc).sum())
        assert np.array_equal(order[b], ro)
        # keep-set as the consumer sees it (detect.py:78-80)
        keep_g = {(i, int(np.argmax(got[b, i]))) for i in range(N) if got[b, i].max() > 0.3}
        keep_r = {(i, int(np.argmax(rc[i]))) for i in range(N) if rc[i].max() > 0.3}
        assert keep_g == keep_r


def test_optimizers_vs_oracle(ops):
    rng = np.random.RandomState(5)
    n = 10007
    w0, g = rng.randn(n).astype(np.float32), rng.randn(n).astype(np.float32)
    s1, s2 = (rng.rand(n) * 0.1).astype(np.float32), (rng.rand(n) * 0.1).astype(np.float32)
    # adam, two steps
    w, m, v = dev(w0), dev(np.zeros(n, np.float32)), dev(np.zeros(n, np.float32))
    rw, rm, rv = w0, np.zeros(n, np.float32), np.zeros(n, np.float32)
    for t in (1, 2):
        alpha = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        ops.adam(w, dev(g), m, v, n, float(alpha), 0.9, 0.999, 1e-8)
        rw, rm, rv = R.adam_step(rw, g, rm, rv, 1e-3, t)
    torch.cuda.synchronize()
    assert_close(host(w), rw, 1e-6, 'adam w')
    assert_close(host(m), rm, 1e-6, 'adam m')
    assert_close(host(v), rv, 1e-6, 'adam v')
    w, a = dev(w0), dev(s1)
    ops.momentum(w, dev(g), a, n, 0.01, 0.9)
    rw, ra = R.momentum_step(w0, g, s1, 0.01, 0.9)
    assert_close(host(w), rw, 1e-6, 'momentum')
    w = dev(w0)
    ops.sgd(w, dev(g), n, 0.01)
    assert_close(host(w), R.gd_step(w0, g, 0.01), 1e-6, 'sgd')
    w, a, b = dev(w0), dev(s1), dev(s2)
    ops.rmsprop(w, dev(g), a, b, n, 0.01, 0.9, 0.5, 1e-10)
    rw, _, _ = R.rmsprop_step(w0, g, s1, s2, 0.01, 0.9, 0.5, 1e-10)
    assert_close(host(w), rw, 1e-5, 'rmsprop')
    w, a = dev(w0), dev(s1 + 0.1)
    ops.adagrad(w, dev(g), a, n, 0.01)
    rw, _ = R.adagrad_step(w0, g, s1 + 0.1, 0.01)
    assert_close(host(w), rw, 1e-6, 'adagrad')
    w, a, b = dev(w0), dev(s1), dev(s2)
    ops.adadelta(w, dev(g), a, b, n, 1.0, 0.95, 1e-8)
    rw, _, _ = R.adadelta_step(w0, g, s1, s2, 1.0, 0.95, 1e-8)
    assert_close(host(w), rw, 1e-5, 'adadelta')
    # ftrl (tf.train.FtrlOptimizer, reference train.py:78): sqrt form, general power, l1 shrinkage to exact zeros; two steps
    for power, l1, l2 in ((-0.5, 0.0, 0.0), (-0.5, 0.05, 0.01), (-0.7, 0.02, 0.0)):
        w, a, b = dev(w0), dev(np.full(n, 0.1, np.float32)), dev(np.zeros(n, np.float32))
        rw, ra, rl = w0, np.full(n, 0.1, np.float32), np.zeros(n, np.float32)
        for step in range(2):
            gs = (g * (1 + step)).astype(np.float32)
            ops.ftrl(w, dev(gs), a, b, n, 0.05, power, l1, l2)
            rw, ra, rl = R.ftrl_step(rw, gs, ra, rl, 0.05, power, l1, l2)
        torch.cuda.synchronize()
        assert_close(host(w), rw, 2e-6 if power == -0.5 else 2e-5, 'ftrl w %s' % ((power, l1, l2),))
        assert_close(host(a), ra, 1e-6, 'ftrl accum')
        assert_close(host(b), rl, 2e-6 if power == -0.5 else 2e-5, 'ftrl linear')
        assert np.array_equal(host(w) == 0, rw == 0)
    x = dev(g)
    ops.scale(x, n, 0.125)
    assert np.array_equal(host(x), g * np.float32(0.125))
    x = dev(g)
    ops.zero_ranges(x, [(0, 8), (100, 100), (5000, n)])
    ref = g.copy(); ref[0:8] = 0; ref[5000:] = 0
    assert np.array_equal(host(x), ref)
    # per-tensor clip_by_norm
    seg = np.array([0, 100, 5000, n], np.int64)
    gd = dev(g)
    ops.clip_by_norm(gd, torch.from_numpy(seg).cuda(), 3, 5.0, torch.zeros(3, dtype=torch.float64, device='cuda'))
    torch.cuda.synchronize()
    ref = np.concatenate([R.clip_by_norm(g[seg[i]:seg[i + 1]], 5.0) for i in range(3)])
    assert_close(host(gd), ref, 1e-6, 'clip_by_norm')


def test_optimizer_slice_updates_equal_whole_arena_update(ops):
    """Optimizer.apply(lo, hi): updating the arena bucket by bucket (data-parallel path) is bit-identical to one launch."""
    from yolo_tf_amd.optim import Optimizer
    n = 10007 * 4
    rng = np.random.RandomState(0)
    p0, g0 = rng.randn(n).astype(np.float32), rng.randn(n).astype(np.float32)
    outs = []
    for cuts in ([0, n], [0, 4096, 20000, 20004, n]):
        opt = Optimizer('adam', None, n, torch.device('cuda'))
        p, g = dev(p0.copy()), dev(g0.copy())
        for t in (1, 2, 3):
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                opt.apply(p, g, 1e-3, t, 0.5, lo, hi)
        torch.cuda.synchronize()
        outs.append(host(p))
    np.testing.assert_array_equal(outs[0], outs[1])



"""Host-side planning logic that decides which kernels a training step launches, checked without a GPU: the partial-row buffer pool of
the engine (who clears what, so that no memset launch is needed), the shapes the folded-finalisation consumers accept, and the
filter-gradient plans that decide which gradient ranges the step has to clear."""
import numpy as np
import pytest
import torch


def test_part_pool_hands_out_clean_buffers_and_has_consumers_clear_the_previous_one():
    from yolo_tf_amd.engine import _PartPool
    pool = _PartPool(3, 1024, 'cpu')
    a = pool.acquire(512)
    assert pool.busy[a] and pool.dirty[a] == 512
    pool.bufs[a][:512] = 1.0                                   # the producer's epilogue writes partial rows
    zbuf, zn = pool.take_to_zero(a)                            # its consumer: nothing else is dirty yet
    assert zbuf is None and zn == 0
    pool.consumed(a)                                           # read, not cleared
    b = pool.acquire(256)
    assert b != a and float(pool.bufs[b].abs().sum()) == 0.0   # a dirty buffer is never handed to a producer
    pool.bufs[b][:256] = 2.0
    zbuf, zn = pool.take_to_zero(b)                            # b's consumer clears a on the side
    assert zbuf is pool.bufs[a] and zn == 512 and pool.dirty[a] == 0
    zbuf[:zn] = 0.0                                            # (what grid_zero does inside the consumer kernel)
    pool.consumed(b)
    # steady state of a step: every producer finds a clean buffer without the fallback memset
    for _ in range(50):
        i = pool.acquire(128)
        assert float(pool.bufs[i].abs().sum()) == 0.0
        pool.bufs[i][:128] = 3.0
        z, n = pool.take_to_zero(i)
        if z is not None:
            z[:n] = 0.0
        pool.consumed(i)
    # a consumer that clears its own rows (the two-launch finalisation) returns the buffer clean
    i = pool.acquire(64)
    pool.bufs[i][:64] = 0.0
    pool.consumed(i, cleared=True)
    assert pool.dirty[i] == 0 and not pool.busy[i]


def test_part_pool_falls_back_to_a_clear_when_every_buffer_is_dirty():
    from yolo_tf_amd.engine import _PartPool
    pool = _PartPool(2, 64, 'cpu')
    for _ in range(2):
        i = pool.acquire(64)
        pool.bufs[i][:] = 5.0
        pool.consumed(i)                                       # consumed, nobody cleared it
    i = pool.acquire(32)                                       # (rare path: costs a memset launch on the device)
    assert float(pool.bufs[i].abs().sum()) == 0.0 and pool.dirty[i] == 32


@pytest.fixture(scope='module')
def q():
    from yolo_tf_amd import _lib
    return _lib.query


def test_folded_finalisation_shape_rule(q):
    """rows x (channel slice <= 128 bf16 / 64 f32) x 8 bytes <= 128 KB per workgroup prologue; 16-byte lane groups must divide 256 threads."""
    F32, BF16 = 0, 1
    assert q('yolo2_bn_fin_supported', 128, 1024, BF16) == 1 and q('yolo2_bn_fin_supported', 129, 1024, BF16) == 0
    assert q('yolo2_bn_fin_supported', 256, 64, BF16) == 1 and q('yolo2_bn_fin_supported', 512, 32, BF16) == 1
    assert q('yolo2_bn_fin_supported', 256, 512, F32) == 1 and q('yolo2_bn_fin_supported', 257, 512, F32) == 0
    assert q('yolo2_bn_fin_supported', 44, 1024, BF16) == 1                      # batch 8, 13x13 stage
    assert q('yolo2_bn_fin_supported', 172, 512, BF16) == 0                      # what batch 8 / 26x26 used to leave (producers now wrap to 16 rows)
    assert q('yolo2_bn_fin_supported', 16, 24, BF16) == 0                        # 3 lane groups do not divide 256
    assert q('yolo2_bn_fin_supported', 0, 64, BF16) == 0 and q('yolo2_bn_fin_supported', 8, 4, BF16) == 0
    from yolo_tf_amd import ops
    assert ops.bn_fin_rows_limit(1024, torch.bfloat16) == 128 and ops.bn_fin_rows_limit(64, torch.bfloat16) == 256
    assert ops.bn_fin_rows_limit(32, torch.bfloat16) == 512 and ops.bn_fin_rows_limit(24, torch.bfloat16) == 0


@pytest.mark.parametrize('batch', [4, 8, 16, 32])
def test_filter_gradient_plans_decide_which_ranges_the_step_clears(q, batch):
    """yolo2_conv2d_wgrad_accumulates: layers whose tile grid covers the chip take ONE pixel range and store (their gradient range needs
    no clearing); the others split the pixels and add atomically into a zeroed range.  The decision depends on the tile count only."""
    BF16 = 1
    def acc(H, cin, cout, k):
        return q('yolo2_conv2d_wgrad_accumulates', batch, H, H, cin, (cin + 7) // 8 * 8, cout, (cout + 7) // 8 * 8, k, BF16)
    assert acc(13, 512, 1024, 3) == 0 and acc(13, 1024, 1024, 3) == 0 and acc(13, 3072, 1024, 3) == 0      # 288 / 576 / 1728 tiles of 128 x 128
    assert acc(26, 256, 512, 3) == 1 and acc(52, 128, 256, 3) == 1 and acc(104, 64, 128, 3) == 1 and acc(208, 32, 64, 3) == 1
    assert acc(13, 1024, 512, 1) == 1 and acc(13, 1024, 125, 1) == 1 and acc(26, 512, 64, 1) == 1
    assert q('yolo2_conv2d_wgrad_accumulates', 0, 13, 13, 8, 8, 8, 8, 3, BF16) == 1                          # bad arguments: be safe, clear


def test_conv1_filter_gradient_plan_one_workgroup_stores_several_add(q):
    """conv_wgrad_c32.hip (32 -> 64 channels, bf16, rows that are a multiple of 16 wide and fit its LDS ring): one workgroup per run of image rows;
    a launch of ONE workgroup owns dW and stores (no clearing), several add atomically.  Shapes it does not take fall to the per-tap kernel's
    plan (f32, channel strides other than 32 / 64, widths that are not a multiple of 16), which splits the pixels of a layer this size."""
    BF16, F32 = 1, 0
    def acc(B, H, W, dtype=BF16, ldx=32, ldy=64):
        return q('yolo2_conv2d_wgrad_accumulates', B, H, W, 32, ldx, 64, ldy, 3, dtype)
    assert acc(1, 1, 16) == 0 and acc(1, 1, 304) == 0                      # one image row: one workgroup
    assert acc(1, 2, 16) == 1 and acc(16, 208, 208) == 1 and acc(8, 304, 304) == 1
    assert acc(16, 208, 208, F32) == 1 and acc(16, 208, 208, BF16, 40, 64) == 1 and acc(16, 200, 200) == 1      # per-tap plan at the layer's size: 208 pixel ranges


def test_arena_layout_alignment_and_bucket_bounds():
    """Every variable starts on a 256-byte boundary (a 425-element bias once shifted every later filter off the 128-byte lines) and
    gradient buckets end on variable boundaries."""
    from yolo_tf_amd.engine import ARENA_ALIGN
    from yolo_tf_amd.parallel import make_buckets
    sizes = [1024 * 425, 425, 3 * 3 * 3072 * 1024, 1024, 1024, 7, 3 * 3 * 32 * 64]
    offs, off = [], 0
    for n in sizes:
        offs.append((off, n))
        off += (n + ARENA_ALIGN - 1) // ARENA_ALIGN * ARENA_ALIGN
    assert all(o % ARENA_ALIGN == 0 for o, _ in offs) and ARENA_ALIGN * 4 == 256
    buckets = make_buckets(offs, off, (8 << 20) // 4)
    starts = {o for o, _ in offs} | {off}
    assert buckets[0][0] == 0 and buckets[-1][1] == off
    assert all(s in starts and e in starts and e > s for s, e in buckets)
    assert all(b[1] == n[0] for b, n in zip(buckets, buckets[1:]))


def test_checkpoint_restore_remaps_optimizer_slots_when_only_the_arena_offsets_moved(tmp_path):
    """A checkpoint written with another arena alignment (same variables, other offsets) keeps its Adam moments: they are moved
    variable by variable.  A different variable set still starts fresh."""
    import numpy as np
    import torch
    from yolo_tf_amd import checkpoint

    class Eng(object):
        def __init__(self, offsets):
            self.param_offsets = offsets
            self.vals = {}

        def get_variables(self):
            return self.vals

        def set_variables(self, values, strict=True):
            self.vals = dict(values)

    class Opt(object):
        name = 'adam'

        def __init__(self, n):
            self.slots = [torch.zeros(n), torch.zeros(n)]

    class Sess(object):
        def __init__(self, offsets, n):
            self.engine, self.optimizer, self.global_step = Eng(offsets), Opt(n), 0

    old = Sess({'a/w': (0, 5), 'b/w': (8, 3)}, 12)          # 4-element alignment
    old.engine.vals = {'a/w': np.arange(5, dtype=np.float32), 'b/w': np.ones(3, np.float32)}
    old.optimizer.slots[0][:] = torch.arange(12.0)
    old.optimizer.slots[1][:] = torch.arange(12.0) * 10
    old.global_step = 7
    path = checkpoint.save(str(tmp_path), old)
    new = Sess({'a/w': (0, 5), 'b/w': (64, 3)}, 80)         # 64-element alignment
    assert checkpoint.restore(path, new) == 7 and new.global_step == 7
    for k, slot in enumerate(new.optimizer.slots):
        scale = 10 ** k
        assert slot[0:5].tolist() == [float(i * scale) for i in range(5)] and slot[64:67].tolist() == [float(i * scale) for i in (8, 9, 10)]
        assert float(slot[5:64].abs().sum()) == 0.0
    other = Sess({'a/w': (0, 5), 'c/w': (64, 3)}, 80)
    other.optimizer.slots[0][:] = 1.0
    checkpoint.restore(path, other)
    assert float(other.optimizer.slots[0].sum()) == 80.0   # untouched: layouts name different variables


def _streamk_segments(tiles, nk, G):
    """The flat (tile, K step) partition of conv_pp.hip / conv_igemm.hip's stream-K launches, restated: workgroup w takes
    [w S / G, (w + 1) S / G) of S = tiles * nk steps and walks it tile by tile -> per workgroup a list of (tile, kt_beg, kt_end)."""
    S = tiles * nk
    out = []
    for w in range(G):
        su, end, segs = w * S // G, (w + 1) * S // G, []
        while su < end:
            t = su // nk
            kb = su - t * nk
            ke = min(nk, kb + (end - su))
            segs.append((t, kb, ke))
            su += ke - kb
        out.append(segs)
    return out


@pytest.mark.parametrize('tiles,nk,G', [(88, 432, 256), (88, 144, 256), (44, 144, 256), (264, 144, 256), (172, 36, 256), (88, 72, 224), (48, 432, 256),
                                        (24, 144, 256), (1, 9, 9), (2, 18, 36), (5, 45, 225)])
def test_stream_k_partition_invariants_the_hand_off_relies_on(tiles, nk, G):
    """What the stream-K hand-off assumes of the partition: every K step belongs to exactly one workgroup; a tail (a segment
    that does not start its tile) is always the FIRST segment of its workgroup, so it is parked before the workgroup can wait for anything;
    the owner of a tile (holder of K step 0) has the lowest index of the tile's workgroups and its partners are the workgroups right after it;
    and -- the case a forced grid on a tiny problem violated, which hung an owner on a flag nobody raises -- no workgroup is empty as long as
    the launch has at most as many workgroups as K steps (launch_conv clamps the grid to that)."""
    assert G <= tiles * nk
    segs = _streamk_segments(tiles, nk, G)
    cover = np.zeros((tiles, nk), np.int32)
    holders = [[] for _ in range(tiles)]
    for w, ss in enumerate(segs):
        assert ss, 'workgroup %d has no K step' % w
        for i, (t, kb, ke) in enumerate(ss):
            cover[t, kb:ke] += 1
            holders[t].append((kb, w))
            if kb > 0:
                assert i == 0, 'a tail segment must be the first segment of its workgroup'
    assert np.all(cover == 1)
    for t in range(tiles):
        hs = sorted(holders[t])
        ws = [w for _, w in hs]
        assert hs[0][0] == 0 and ws == list(range(ws[0], ws[0] + len(ws))), 'owner first, partners consecutive'


def test_stream_k_grid_larger_than_the_k_steps_leaves_workgroups_empty():
    """Why launch_conv clamps: 9 K steps over 256 workgroups -- the owner of the only tile (workgroup 28) would wait for workgroup 29's flag,
    and workgroup 29 has nothing to park."""
    segs = _streamk_segments(1, 9, 256)
    assert [w for w, s in enumerate(segs) if s][:2] == [28, 56] and not segs[29]


# ---- row-of-taps filter-gradient kernel (csrc/conv_wgrad3.hip): host-side pieces checked without a GPU
def _magic(q, d):
    import ctypes
    m, s = ctypes.c_uint(0), ctypes.c_uint(0)
    assert q('yolo2_debug_magic_u32', d, ctypes.byref(m), ctypes.byref(s)) == 0
    return int(m.value), int(s.value)


@pytest.mark.parametrize('d', [2, 3, 4, 5, 7, 8, 11, 13, 14, 16, 20, 21, 27, 32, 39, 53, 64, 77, 105, 128, 209, 256, 417, 609, 1000, 65535])
def test_magic_division_is_exact_below_2_to_31(q, d):
    """floor(x / d) == (x * m >> 32) >> s: every x below 2^24 (the kernel's padded positions are), the boundaries up to 2^31 and random values."""
    m, s = _magic(q, d)
    assert m < 2 ** 32
    x = np.arange(0, 1 << 24, dtype=np.uint64)
    assert np.array_equal(((x * np.uint64(m)) >> np.uint64(32)) >> np.uint64(s), x // np.uint64(d))
    rng = np.random.RandomState(d)
    edge = np.concatenate([np.arange(1, 4000, dtype=np.uint64) * np.uint64(d) - np.uint64(1), np.arange(1, 4000, dtype=np.uint64) * np.uint64(d),
                           (np.uint64(2 ** 31 - 1) - np.arange(0, 4000, dtype=np.uint64)), rng.randint(0, 2 ** 31, 200000).astype(np.uint64)])
    edge = edge[edge < np.uint64(2 ** 31)]
    assert np.array_equal(((edge * np.uint64(m)) >> np.uint64(32)) >> np.uint64(s), edge // np.uint64(d))


@pytest.mark.parametrize('B,H,W,Cin,Cout', [(2, 13, 13, 8, 8), (3, 5, 7, 4, 6), (1, 2, 2, 3, 2), (2, 6, 15, 5, 3)])
def test_padded_pixel_index_decomposition_equals_the_filter_gradient(q, B, H, W, Cin, Cout):
    """The kernel's algebra, restated with NumPy: over the padded index q (row pitch W + 1, one zero column per image row), with X rows
    taken dh image rows away through the source address (zeros outside the image) and the three taps of a kernel row read at q - 1, q, q + 1,
    sum_q Xs[q + dw] dYp[q] is the SAME-padded 3x3 filter gradient -- no (pixel, tap) mask anywhere.  The source decomposition uses the
    library's magic constants the way the kernel's DMA address arithmetic does."""
    from oracle import yolo2_ref as R
    rng = np.random.RandomState(B * 100 + H * 10 + W)
    x = rng.randn(B, H, W, Cin).astype(np.float32)
    dy = rng.randn(B, H, W, Cout).astype(np.float32)
    ref = R.conv2d_wgrad(x, dy, 3, 3)
    mW, sW = _magic(q, W + 1)
    mH, sH = _magic(q, H)
    Mp, halo = B * H * (W + 1), 4
    xf, dyf = x.reshape(B * H * W, Cin), dy.reshape(B * H * W, Cout)
    got = np.zeros((3, 3, Cin, Cout), np.float64)
    for dh in (-1, 0, 1):
        # what the DMA stages: X rows for padded positions -halo .. Mp + halo, dY rows for 0 .. Mp
        xs = np.zeros((Mp + 2 * halo, Cin))
        for i in range(Mp + 2 * halo):
            qq = (i - halo) & 0xffffffff                                   # unsigned wrap below zero, as in the kernel
            Rr = ((qq * mW) >> 32) >> sW
            c = (qq - Rr * (W + 1)) & 0xffffffff
            r = Rr - (((Rr * mH) >> 32) >> sH) * H
            if qq < Mp and c < W and 0 <= r + dh < H:
                xs[i] = xf[qq - Rr + dh * W]
        dyp = np.zeros((Mp, Cout))
        for qq in range(Mp):
            Rr = ((qq * mW) >> 32) >> sW
            c = qq - Rr * (W + 1)
            if c < W:
                dyp[qq] = dyf[qq - Rr]
        for dw in (-1, 0, 1):
            got[dh + 1, dw + 1] = xs[halo + dw:halo + dw + Mp].T @ dyp
    assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()


def test_row_kernel_plans(q):
    """Variant and pixel-range rule of the row-of-taps kernel on a 256-CU device: 13x13 stages store (one range, plain stores, L2-stationary
    order), the split layers keep one workgroup per CU at most with ranges in multiples of 8 (one XCD per range), every range is whole ring
    turns, ranges cover the padded pixels; narrow layers stay on the per-tap kernel."""
    import ctypes
    def plan(B, H, cin, cout, cus=256, force=-1):
        out = (ctypes.c_int * 9)()
        assert q('yolo2_debug_wgrad_row_plan', B, H, H, cin, cout, cus, force, out) == 0
        return dict(zip(('variant', 'ranges', 'qchunk', 'blocks', 'remap', 'direct', 'BC', 'BN', 'waves'), list(out)))
    for B in (4, 8, 16, 32):
        for H, cin, cout, v in ((13, 512, 1024, 2), (13, 1024, 1024, 4), (13, 3072, 1024, 2)):
            p = plan(B, H, cin, cout)
            assert (p['variant'], p['ranges'], p['direct'], p['remap']) == (v, 1, 1, 2), p
            assert p['blocks'] == 3 * (cin // p['BC']) * (cout // p['BN']) and p['qchunk'] >= B * H * (H + 1)
        for H, cin, cout in ((26, 256, 512), (52, 128, 256), (104, 64, 128)):
            p = plan(B, H, cin, cout)
            step = 4 * 64
            assert (p['variant'], p['direct'], p['BC'], p['BN']) == (5, 0, 64, 64) and p['ranges'] >= 2, p
            assert p['qchunk'] % step == 0 and p['qchunk'] >= 4 * step
            assert (p['ranges'] - 1) * p['qchunk'] < B * H * (H + 1) <= p['ranges'] * p['qchunk']
            cols = 3 * -(-cin // p['BC']) * -(-cout // p['BN'])
            assert cols * p['ranges'] <= 256 and p['blocks'] >= cols * p['ranges']
            assert p['remap'] == (1 if p['ranges'] >= 8 else 0)
        assert plan(B, 208, 32, 64)['variant'] == -1             # <= 32 input channels: the per-tap kernel's tap-pair form
    assert plan(16, 52, 128, 256)['ranges'] == 8                 # 256 / 24 = 10 -> 8: one range per XCD
    assert plan(16, 416, 3, 32)['variant'] == -1                 # the image layer keeps its own kernel
    assert plan(16, 1, 64, 64)['variant'] == -1                  # H = 1: no division constant for it
    assert plan(64, 608, 64, 64)['variant'] == -1                # padded positions beyond the 24-bit multiplies
    assert plan(16, 13, 512, 1024, force=0)['variant'] == -1     # experiment variants are not in the product build

"""Executes a ``graph.Graph`` on one MI355X through the C-ABI kernels.

Plays the role of the TF-1 session + tf.gradients + optimizer.apply_gradients in the reference
(train.py:109-145 builds them; slim.learning.train runs ``sess.run(train_op)``): it owns the
device buffers (flat f32 parameter / gradient / optimizer-state arenas, activation and
activation-gradient buffers in the compute dtype, prepared filter layouts) and launches the HIP
kernels for forward, the YOLOv2 loss, backward and the update on the current HIP stream.

Memory layout decisions (288 GB HBM: nothing is recomputed, nothing is re-allocated per step):
  * all trainable variables live in ONE flat f32 buffer in REVERSE creation order, gradients in a
    twin buffer: the optimizer is one launch, and the data-parallel all-reduce works on contiguous
    ranges that complete front-to-back as backward proceeds (parallel.GradReducer).
  * activations are NHWC with a pixel stride padded to 8 channels; tf.concat operands alias channel
    slices of the concat buffer (producers write in place, no copy kernel).
"""
import numpy as np
import os

import torch

from . import ops, _lib
from .graph import pad8

BN_EPS = 1e-5        # slim.batch_norm(epsilon=1e-5)   model/yolo2/inference.py:63
BN_DECAY = 0.999     # [TF-sem] slim.batch_norm default decay
LEAKY_ALPHA = 0.1    # model/yolo/function.py:21

_DTYPES = {'bf16': torch.bfloat16, 'bfloat16': torch.bfloat16, 'f32': torch.float32, 'float32': torch.float32}


BN_PART_ROWS = 256           # YOLO2_BN_PART_ROWS of include/yolo2_hip.h: rows per plane of a partial-sum buffer
ARENA_ALIGN = 64      # elements: every variable starts on a 256-byte boundary


def layout_params(graph):
    """Flat f32 arena layout of the trainable variables: REVERSE creation order (the last layer first), every
    variable 256-byte aligned (16 bytes would do for the vector accesses, but a 425-element bias then shifts every later filter off
    the 128-byte lines: the COCO-80 head made the Adam pass 18 % and the atomic filter gradients 20-25 % slower than VOC-20's,
    profiles/r03_arena_alignment.txt).  Returns ({name: (offset, size)}, total elements).  Pure host logic."""
    offsets, off = {}, 0
    for v in reversed(graph.trainable()):
        offsets[v.name] = (off, v.size)
        off += (v.size + ARENA_ALIGN - 1) // ARENA_ALIGN * ARENA_ALIGN
    return offsets, off


def layer_end_offsets(graph, offsets):
    """For every conv op: the arena offset below which all gradients are final once that op's backward has
    run (backward visits ops in reverse order, so these ends grow monotonically)."""
    ends = {}
    for op in graph.ops:
        if op['kind'] == 'conv':
            names = [op['weights'].name] + [op[k].name for k in ('gamma', 'beta', 'biases') if k in op]
            ends[op['name']] = max(offsets[n][0] + (offsets[n][1] + ARENA_ALIGN - 1) // ARENA_ALIGN * ARENA_ALIGN for n in names)
    return ends


_SKEW_KB = 20        # (profiles/r03_arena_stagger.txt)
_skew_counter = [0]


def staggered(n, dtype, device, fill=0.0):
    """A device buffer of ``n`` elements whose start is offset by (k mod 16) x 20 KiB from its allocation, k counting the big buffers of the
    process.  The caching allocator hands out 2 MiB-aligned blocks, so equally sized arenas that one kernel walks in lock step (Adam reads
    element i of four of them at once; a filter gradient reads pixel m of x and dY) otherwise sit at identical offsets of the HBM
    channel interleave: measured 418 us for the Adam pass with 2 MiB-aligned arenas against 377 us with 4 KiB of stagger; whole step +0.5..0.8 % with 4-68 KiB, best at 20
    (scripts/arena_alias_bench.py, profiles/r03_arena_stagger.txt)."""
    esz = torch.empty(0, dtype=dtype).element_size()
    if _SKEW_KB <= 0 or n * esz < (1 << 20):
        return torch.full((n,), fill, dtype=dtype, device=device) if fill else torch.zeros(n, dtype=dtype, device=device)
    k = _skew_counter[0] = _skew_counter[0] + 1
    skew = (k % 16) * _SKEW_KB * 1024 // esz
    base = torch.full((n + 16 * _SKEW_KB * 1024 // esz,), fill, dtype=dtype, device=device) if fill else torch.zeros(n + 16 * _SKEW_KB * 1024 // esz, dtype=dtype, device=device)
    return base[skew:skew + n]


class _PartPool(object):
    """Host-side bookkeeping of the partial-row buffers (stream order = call order on the main stream).  A producer takes a CLEAN
    buffer; its consumer marks it USED and hands one earlier USED buffer to its own kernel to clear (``take_to_zero``)."""

    def __init__(self, n, floats, device):
        self.bufs = [torch.zeros(floats, dtype=torch.float32, device=device) for _ in range(n)]
        self.dirty = [0] * n          # floats from the start of the buffer that may be non-zero; 0 = clean
        self.busy = [False] * n       # produced, not consumed yet

    def acquire(self, floats):
        """-> index of a clean buffer, now owned by a producer that may dirty ``floats`` floats of it."""
        for i, (d, b) in enumerate(zip(self.dirty, self.busy)):
            if d == 0 and not b:
                break
        else:       # every buffer is waiting for its consumer or for a clear: clear one that has been consumed (rare; costs a launch)
            i = next(j for j, b in enumerate(self.busy) if not b)
            self.bufs[i][:self.dirty[i]].zero_()
        self.dirty[i] = int(floats)
        self.busy[i] = True
        return i

    def consumed(self, i, cleared=False):
        self.busy[i] = False
        if cleared:
            self.dirty[i] = 0

    def take_to_zero(self, exclude):
        """-> (buffer, floats) of one consumed dirty buffer for the calling consumer's kernel to clear, or (None, 0)."""
        for i, (d, b) in enumerate(zip(self.dirty, self.busy)):
            if i != exclude and d and not b:
                self.dirty[i] = 0
                return self.bufs[i], d
        return None, 0


class Engine(object):
    def __init__(self, graph, batch_size, dtype='bf16', training=True, seed=0, device=None, sync_bn=False, side_priority=-1):
        if not torch.cuda.is_available():
            raise RuntimeError('yolo_tf_amd.Engine needs an MI355X (no CPU path exists)')
        ops._lib.load()
        self.side_priority = side_priority
        # Synchronised batch normalisation (data-parallel option, [mi355x] sync_bn): batch moments and the BN-backward sums are summed over
        # the replicas, so N ranks x B images train like one process with N x B images.  The reference is single-device (local statistics
        # are what N independent replicas of it would compute; that stays the default).  Costs two small collectives per BN layer and
        # step, and takes the two-launch finalisation forms (the sums have to exist outside a kernel to be exchanged).
        self.sync_bn = bool(sync_bn) and training
        self.bn_group, self.bn_world = None, 1        # set by TrainSession once the process group exists
        self.graph = graph
        self.B = int(batch_size)
        self.dtype = _DTYPES[dtype] if isinstance(dtype, str) else dtype
        self.training = training
        self.device = torch.device(device if device is not None else 'cuda:%d' % torch.cuda.current_device())
        self._alloc_variables()
        self._alloc_activations()
        self.init_variables(seed)
        self._filters_dirty = True
        self._zero_ranges = None
        self.reg_loss = torch.zeros(1, dtype=torch.float64, device=self.device)      # sum of the regularisation terms of the last backward
        self._has_l2 = any(op.get('l2') for op in graph.ops)                         # (only the YOLO v1 family's fully connected layers)
        self.dropout_masks = None    # {op name: uint8 mask}: fixed keep masks (parity tests); None = drawn on the device per step
        self.dropout_seed = int(seed) + 1
        self.dropout_rank = 0        # data-parallel rank and global step (set by TrainSession): every replica draws its own masks, and a
        self.dropout_step = 0        # resumed run continues the mask sequence instead of replaying it from the start
        self._dropout_calls = 0
        self._masks = {}
        self._phase = 'fwd'          # 'fwd' | 'dgrad': which sweep a conv launch belongs to (timer tag)
        self.kernel_timer = None     # optional bench.KernelTimer: HIP events around the dominant conv kernel

    # ---------------------------------------------------------------- variables
    def _alloc_variables(self):
        g = self.graph
        other = [v for v in g.variables.values() if not v.trainable]
        self.param_offsets, off = layout_params(g)
        self.n_params = off
        self.params = staggered(off, torch.float32, self.device)
        self.grads = staggered(off, torch.float32, self.device) if self.training else None
        self.state_offsets = {}
        off = 0
        for v in other:
            self.state_offsets[v.name] = (off, v.size)
            off += (v.size + 3) // 4 * 4
        self.state = torch.zeros(max(off, 4), dtype=torch.float32, device=self.device)
        # Snapshot of the non-trainable state taken at the start of every training forward.  The convolution epilogues and the
        # consumers that finish the batch statistics (yolo2_bn_leaky_fin & co.) shift their sums by svar[moving_mean]: reading the live
        # moving mean would race with the workgroup that updates it in place (thousands of workgroups read the shift after their
        # prologue; the first one of each channel slice writes the moving average).
        self.state_snap = torch.zeros_like(self.state)
        self.svar = {}
        self.var = {}
        self.gvar = {}
        for name, (o, n) in self.param_offsets.items():
            self.var[name] = self.params[o:o + n]
            if self.training:
                self.gvar[name] = self.grads[o:o + n]
        for name, (o, n) in self.state_offsets.items():
            self.var[name] = self.state[o:o + n]
            self.svar[name] = self.state_snap[o:o + n]
        # per-tensor segments (for clip_by_norm) in buffer order
        segs = sorted(self.param_offsets.values())
        self.seg_off = torch.tensor([s[0] for s in segs] + [self.n_params], dtype=torch.int64, device=self.device)
        self.n_seg = len(segs)

    def init_variables(self, seed=0):
        """[TF-sem] slim initialisers, seeded (the reference parses --seed but never uses it, train.py:161)."""
        rng = np.random.RandomState(seed)
        for v in self.graph.variables.values():
            self.var[v.name].copy_(torch.from_numpy(v.init(rng, v.shape).reshape(-1)))
        self._filters_dirty = True

    def get_variables(self):
        torch.cuda.synchronize()
        return {v.name: self.var[v.name].cpu().numpy().reshape(v.shape) for v in self.graph.variables.values()}

    def set_variables(self, values, strict=True):
        for v in self.graph.variables.values():
            if v.name in values:
                a = np.asarray(values[v.name], np.float32)
                assert a.shape == v.shape, (v.name, a.shape, v.shape)
                self.var[v.name].copy_(torch.from_numpy(np.ascontiguousarray(a).reshape(-1)))
            elif strict:
                raise KeyError(v.name)
        self._filters_dirty = True

    def get_gradients(self):
        torch.cuda.synchronize()
        return {v.name: self.gvar[v.name].cpu().numpy().reshape(v.shape) for v in self.graph.trainable()}

    # ---------------------------------------------------------------- activations
    def _alloc_activations(self):
        """Root buffers (by tensor NAME, sized for the graph the engine is built with -- the largest input size of a multi-scale
        run) and per-layer state; then the first binding.  Activations are flat [B*h*w][ld] images, so a smaller input size uses a
        prefix of every buffer and the padding lanes (index mod ld >= c) sit at the same flat positions for every size."""
        B, T, dev = self.B, self.dtype, self.device
        self._roots, self._groots = {}, {}
        for t in self.graph.tensors:
            if t.base is None:
                self._roots[t.name] = staggered(B * t.h * t.w * t.ld, T, dev)
                if self.training and t not in self.graph.inputs.values():
                    self._groots[t.name] = staggered(B * t.h * t.w * t.ld, T, dev)
        self.conv = {}
        max_y = 0
        max_c = 8
        for op in self.graph.ops:
            if op['kind'] != 'conv':
                continue
            k, cin, cout = op['ksize'], op['cin'], op['cout']
            cp, ldy = pad8(cin), pad8(cout)
            st = {'Ffwd': staggered(cout * k * k * cp, T, dev)}
            if self.training and op['x'] not in self.graph.inputs.values():
                st['Fdgr'] = staggered(cin * k * k * ldy, T, dev)
            if op['bn']:
                st['mean'] = torch.zeros(cout, dtype=torch.float32, device=dev)
                st['var'] = torch.ones(cout, dtype=torch.float32, device=dev)
            self.conv[op['name']] = st
            max_y = max(max_y, B * op['out'].h * op['out'].w * ldy)
            max_c = max(max_c, ldy)
        self._bindings = {}
        self._tmp_roots = {}
        self.fold_finalize = os.environ.get('YOLO2_FOLD_FINALIZE', '1') != '0' and not self.sync_bn
        self.pool_ymax = self.fold_finalize
        self._bind(self.graph)
        self.fold_bn = os.environ.get('YOLO2_FOLD_BN', '1') != '0'
        self.fuse_bn_stats = os.environ.get('YOLO2_FUSE_BN_STATS', '1') != '0'
        # Partial rows of the fused statistics ([2][YOLO2_BN_PART_ROWS][C] each).  Producers (convolution epilogues) need a zero buffer; the
        # consumer that finalises the rows in its own prologue only reads them and clears a buffer an EARLIER consumer is done with
        # (_PartPool below) -- no finalisation launch, no memset launch.  YOLO2_FOLD_FINALIZE=0: separate finalisation kernels (A/B).
        self.parts = _PartPool(3, 2 * 256 * max(max_c, 8), dev)
        self.bn_part = self.parts.bufs[0]                       # (kept for callers that drive the two-launch form directly)
        # image layer recomputed inside its consumers instead of stored: 'infer' (default: detect only -- batch 256: 12.5 -> 11.3 ms),
        # '1' (training too: measured neutral, the recomputing backward kernels are VALU-bound -- profiles/r03_first_layer_fused.md), '0' never
        self.fuse_first = os.environ.get('YOLO2_FUSE_FIRST', 'infer')
        self.fuse_first_wgrad = os.environ.get('YOLO2_FUSE_FIRST_WGRAD', '1') != '0'
        self._bz_pending = {}                                    # producer layer -> (buffer, rows): BN-backward sums waiting for their apply pass
        self._fin_rows_limit = {}
        # scratch sizes come from the library's own queries (include/yolo2_hip.h yolo2_*_workspace_bytes)
        conv_bytes = 0
        for op in self.graph.ops:
            if op['kind'] == 'conv':
                x = op['x']
                for cp, nf in ((pad8(op['cin']), op['cout']), (pad8(op['cout']), op['cin'])):       # forward, data gradient
                    conv_bytes = max(conv_bytes, ops.workspace_bytes('conv2d', B, x.h, x.w, cp, nf, op['ksize'], ops.dtype_code(T)))
        self.conv_ws = torch.zeros(conv_bytes // 4 + 1024, dtype=torch.float32, device=dev)   # stream-K tile slots / K-sliced partial image
        ws_bytes = max(ops.workspace_bytes('bn', max_c), ops.workspace_bytes('bias_grad', max_c)) + ops.workspace_bytes('image_prep', B)
        self.ws = torch.zeros(ws_bytes // 8 + 64, dtype=torch.float64, device=dev)   # reduction partials
        if self.training:
            # dY scratch ring: the filter gradient of layer L runs on a side stream concurrently with the data gradient
            # (and the following layers' backward) on the main stream, so dY(L) must outlive the next layers' writes
            self.dy_ring = [staggered(max_y, T, dev) for _ in range(3)]
            self.dy_free = [None, None, None]                    # event: the side stream has finished reading that buffer
            # The filter-gradient stream at HIGH priority (YOLO2_SIDE_PRIORITY=0: normal, A/B).  Both streams' big kernels need a whole CU's LDS and
            # share the chip workgroup by workgroup; with equal priority the filter gradients fall behind the dependency chain on the main stream and
            # pile up after its last layer, in front of Adam.  Measured in one call (profiles/r06_new_kernels.txt, last block): 3.470 -> 3.448 ms;
            # the step on its own (non-default) stream of either priority: 3.50 .. 3.53 ms.
            # (side_priority: data-parallel sessions pass 0 -- the collective's stream is the one high-priority stream there, as measured in rounds 3-5)
            prio = int(os.environ.get('YOLO2_SIDE_PRIORITY', str(self.side_priority)))
            lo, hi = torch.cuda.Stream.priority_range()
            self.side_stream = torch.cuda.Stream(device=dev, priority=max(min(prio, max(lo, hi)), min(lo, hi)))
            self.overlap_wgrad = os.environ.get('YOLO2_OVERLAP_WGRAD', '1') != '0'   # 0: single stream (clean per-kernel profiles)
            self.overlap_max_m = 1 << 40   # A/B: overlap only layers with at most this many output pixels
        self.img = None

    def _bind(self, graph):
        """Views of the root buffers for one traced input size + the shape-dependent plans; makes it the current binding."""
        B, T, dev = self.B, self.dtype, self.device
        act, gact = {}, {}
        for t in graph.tensors:
            r, off, ld = t.storage()
            root = self._roots[r.name]
            assert B * r.h * r.w * r.ld <= root.numel(), 'input size exceeds the one the engine was allocated for (%s)' % r.name
            act[t] = (root[off:], ld)
            if r.name in self._groots:
                gact[t] = (self._groots[r.name][off:], ld)
        # BN + leaky + max-pool fusion: a batch-normalised conv whose output feeds one stride-2 pool and nothing else never
        # materialises its full-resolution activation (forward) or that activation's gradient (backward)
        fused_pool, fwd_pool = {}, {}
        if os.environ.get('YOLO2_FUSE_POOL', '1') != '0':
            uses = {}
            for op in graph.ops:
                for t in (op.get('inputs') or [op['x']]):
                    uses[t] = uses.get(t, 0) + 1
            producers = {op['out']: op for op in graph.ops if op['kind'] == 'conv'}
            for op in graph.ops:
                x = op.get('x')
                if (op['kind'] == 'pool' and op['stride'] == 2 and x in producers and producers[x]['bn'] and uses.get(x, 0) == 1
                        and x.h % 2 == 0 and x.w % 2 == 0 and act[x][1] == x.c and act[op['out']][1] == op['out'].c
                        and act[producers[x]['y']][1] == x.c and x.c // (8 if T == torch.bfloat16 else 4) <= 256):
                    fused_pool[x] = op
                    st = self.conv[producers[x]['name']]
                    need = B * (x.h // 2) * (x.w // 2) * x.c
                    if self.training and ('pool_idx' not in st or st['pool_idx'].numel() < need):
                        st['pool_idx'] = torch.zeros(need, dtype=torch.uint8, device=dev)
                    # the raw output AT the arg-max, a quarter of y: all the layer's backward reduction needs
                    if self.training and self.pool_ymax and ('pool_ymax' not in st or st['pool_ymax'].numel() < need):
                        st['pool_ymax'] = staggered(need, T, dev)
            # forward only: the pool of an activation that has OTHER readers too (Darknet-19's 26x26 passthrough) still comes out of the BN
            # pass, which then stores both resolutions (yolo2_bn_leaky_pool_fin, A_full); the backward keeps the separate kernels
            for op in graph.ops:
                x = op.get('x')
                if (self.training and op['kind'] == 'pool' and op['stride'] == 2 and x in producers and producers[x]['bn'] and x not in fused_pool
                        and uses.get(x, 0) > 1 and x.h % 2 == 0 and x.w % 2 == 0 and act[x][1] == x.c and act[op['out']][1] == op['out'].c
                        and act[producers[x]['y']][1] == x.c and x.c // (8 if T == torch.bfloat16 else 4) <= 256
                        ):
                    fwd_pool[x] = op
        # BN-backward sums in the consumer's data-gradient epilogue: a batch-normalised conv whose full-resolution activation has
        # exactly one reader, a convolution writing that activation's gradient directly (no concat slice, no fan-out)
        bn_bwd_fused = {}
        if self.training and os.environ.get('YOLO2_FUSE_BN_BWD', '1') != '0':
            uses = {}
            for op in graph.ops:
                for t in (op.get('inputs') or [op['x']]):
                    uses[t] = uses.get(t, 0) + 1
            producers = {op['out']: op for op in graph.ops if op['kind'] == 'conv'}
            for op in graph.ops:
                x = op.get('x')
                # (measured per layer, profiles/r03_bn_bwd_fusion_per_layer.txt: in a single-stream trace the epilogue sums win 3-10 us per
                # launch up to 26x26 at batch 16 and lose 1-13 us on the 52x52 / 104x104 layers; with the filter gradients overlapped on the
                # side stream the whole step is equal either way, so every qualifying layer stays fused)
                # (round 6: the library's launch rule is asked first -- conv4's data gradient at batch 16 runs plain, its producer's sums come
                # from the reduction pass that feeds the folded apply pass like every un-fused layer's)
                if (op['kind'] == 'conv' and x in producers and producers[x]['bn'] and 'fold_bias' not in self.conv[producers[x]['name']]
                        and uses.get(x, 0) == 1 and x not in fused_pool and x in gact and gact[x][1] == x.c and act[producers[x]['y']][1] == x.c
                        and ops.conv2d_dgrad_bn_fuses(self.B, x.h, x.w, op['cin'], op['ksize'], self.dtype)):
                    bn_bwd_fused[op['name']] = producers[x]
        inp = next(iter(graph.inputs.values()))
        self._bindings[(inp.h, inp.w)] = {'graph': graph, 'act': act, 'gact': gact, 'fused_pool': fused_pool, 'zero_ranges': None, 'tmp_grad': {},
                                          'bn_bwd_fused': bn_bwd_fused, 'fwd_pool': fwd_pool}
        self._use(inp.h, inp.w)

    def _use(self, h, w):
        bnd = self._bindings[(h, w)]
        self._cur = bnd
        self.graph, self.act, self.gact, self.fused_pool = bnd['graph'], bnd['act'], bnd['gact'], bnd['fused_pool']
        self._zero_ranges, self.tmp_grad = bnd['zero_ranges'], bnd['tmp_grad']
        self.bn_bwd_fused, self.fwd_pool = bnd['bn_bwd_fused'], bnd['fwd_pool']

    def add_size(self, graph):
        """Multi-scale training (BASELINE configs[3]): binds another traced input size of the SAME network (same variables, any
        size not larger than the construction-time one) to the same buffers."""
        names = [v.name for v in graph.variables.values()]
        assert names == [v.name for v in self._bindings[next(iter(self._bindings))]['graph'].variables.values()], 'different network'
        cur = self._cur
        self._bind(graph)
        self._cur = cur
        self._use(*[k for k, v in self._bindings.items() if v is cur][0])

    def set_size(self, height, width):
        """Switches every following forward / backward to a bound input size; parameters, statistics and optimizer state are shared."""
        if (height, width) not in self._bindings:
            raise KeyError('input size %dx%d was not bound (Engine.add_size)' % (height, width))
        self._use(height, width)

    # ---------------------------------------------------------------- helpers
    def _conv(self, P, F, bias, O, H, W, Cp, ldp, Nf, ldo, k, real_k, bn_shift=None, bn_bwd=None):
        """yolo2_conv2d launch; when a timer is attached, the 3x3 launches that take the 128-wide filter tile (Nf > 64:
        conv_igemm_kernel<.., KS = 3, ..>, the kernel that carries 64 % of the training FLOPs; the filter gradient carries
        most of the rest) are bracketed by HIP events; the 1x1 launches are tagged separately.
        ``real_k`` = unpadded reduction length, for the algorithmic FLOP count."""
        t = self.kernel_timer if Nf > 64 else None
        if t is not None:
            t.start(2.0 * self.B * H * W * Nf * real_k, self._phase if k == 3 else '1x1')
        pending = False
        part = None
        if bn_shift is not None:     # training forward of a batch-normalised layer: statistics from the conv epilogue
            part = self.parts.acquire(2 * 256 * Nf)
            ops.conv2d_bn(P, F, O, self.conv_ws, self.B, H, W, Cp, ldp, Nf, ldo, k, bn_shift, self.parts.bufs[part])
        elif bn_bwd is not None:     # data gradient + the producer layer's dgamma / dbeta sums from the same epilogue
            y, mean, var, gamma, beta, dgam, dbet = bn_bwd
            part = self.parts.acquire(2 * 256 * Nf)
            pending = ops.conv2d_dgrad_bn(P, F, O, self.conv_ws, self.B, H, W, Cp, ldp, Nf, ldo, k, y, mean, var, gamma, beta, dgam, dbet,
                                          self.parts.bufs[part], self.ws, BN_EPS, LEAKY_ALPHA)
        else:
            ops.conv2d_ws(P, F, bias, O, self.conv_ws, self.B, H, W, Cp, ldp, Nf, ldo, k)
        if t is not None:
            t.stop()
        if bn_bwd is not None:
            if not pending:          # the two-step form ran inside the call: the buffer was not touched
                self.parts.dirty[part] = 0
                self.parts.consumed(part)
                return None
            rows = ops.last_bn_part_rows()
            if self.fold_finalize and ops.bn_fin_supported(rows, Nf, self.dtype):
                return part, rows    # the producer layer's backward apply pass sums the rows itself
            ops.bn_part_to_grads(self.parts.bufs[part], Nf, bn_bwd[5], bn_bwd[6])      # (reads and clears all rows)
            self.parts.consumed(part, cleared=True)
            return None
        if bn_shift is not None:
            return part, ops.last_bn_part_rows()
        return None

    def _prepare_filters(self):
        """HWIO f32 masters -> both MFMA operand layouts of every layer, one launch (descriptor table built once)."""
        if not self._filters_dirty:
            return
        self._filter_descs()
        for op in self.graph.ops:
            if op['kind'] == 'conv' and 'fold_w' in self.conv[op['name']]:
                st = self.conv[op['name']]
                ops.bn_fold(self.var[op['weights'].name], self.var[op['gamma'].name], self.var[op['beta'].name], self.var[op['moving_mean'].name],
                            self.var[op['moving_variance'].name], st['fold_w'], st['fold_bias'], op['ksize'] ** 2 * op['cin'], op['cout'], BN_EPS)
        ops.filter_prep_batch(self._fdesc, self._fdesc_n, self._fdesc_blocks, self.dtype)
        self._filters_dirty = False

    def _filter_descs(self):
        """Device table of yolo2_filter_desc (one per convolution, built once) + the non-filter parameter ranges of the arena."""
        if getattr(self, '_fdesc', None) is not None:
            return
        from ._lib import FilterDesc
        convs = [op for op in self.graph.ops if op['kind'] == 'conv']
        arr = (FilterDesc * len(convs))()
        first = 0
        for d, op in zip(arr, convs):
            st = self.conv[op['name']]
            k, ldcin, ldcout = op['ksize'], pad8(op['cin']), pad8(op['cout'])
            if self._folds(op):
                # inference: moving statistics folded into the filter (W * gamma/sigma) and a bias (beta - mean*gamma/sigma)
                st['fold_w'] = torch.empty_like(self.var[op['weights'].name])
                st['fold_bias'] = torch.empty(op['cout'], dtype=torch.float32, device=self.device)
            d.W = (st['fold_w'] if 'fold_w' in st else self.var[op['weights'].name]).data_ptr()
            d.Ffwd = st['Ffwd'].data_ptr()
            d.Fdgr = st['Fdgr'].data_ptr() if 'Fdgr' in st else None
            d.ksize, d.cin, d.ldcin, d.cout, d.ldcout, d.first_block = k, op['cin'], ldcin, op['cout'], ldcout, first
            first += ops.filter_prep_blocks(k, ldcin, ldcout)
        raw = bytes(arr)
        self._fdesc = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        self._fdesc_n, self._fdesc_blocks = len(convs), first
        # the same descriptors as one-layer tables (first_block = 0 each): the per-layer optimizer launches of adam_update_layer
        import ctypes
        self._fdesc1_info = {}
        size = ctypes.sizeof(FilterDesc)
        bounds = [d.first_block for d in arr] + [first]
        for i, (d, op) in enumerate(zip(arr, convs)):
            self._fdesc1_info[op['name']] = (i * size, bounds[i + 1] - bounds[i])
            d.first_block = 0
        self._fdesc1 = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        covered = sorted(self.param_offsets[op['weights'].name] for op in self.graph.ops if op['kind'] == 'conv')
        small = sorted(v for k, v in self.param_offsets.items() if v not in covered)
        flat = [x for o, n in small for x in (o, n)]
        self._small = torch.tensor(flat if flat else [0, 0], dtype=torch.int64, device=self.device)
        self._n_small = len(small)

    def adam_update_and_prepare(self, m, v, alpha, beta1, beta2, eps, gscale):
        """Adam over the whole arena + both operand layouts of every filter in ONE launch (yolo2_adam_filter_prep); the next forward
        finds its filters prepared."""
        self._filter_descs()
        ops.adam_filter_prep(self._fdesc, self._fdesc_n, self._fdesc_blocks, self._small, self._n_small, self.params, self.grads, m, v,
                             alpha, beta1, beta2, eps, gscale, self.dtype)
        self._filters_dirty = False

    def adam_update_layer(self, op, m, v, alpha, beta1, beta2, eps, gscale):
        """Adam + both operand layouts of ONE layer's filter, on the current stream (the same kernel as adam_update_and_prepare over a
        one-entry descriptor table: bit-identical).  The caller orders it behind the layer's filter gradient AND its data gradient -- the
        launch rewrites the data-gradient operand Fdgr -- and finishes the step with adam_update_small."""
        self._filter_descs()
        off, blocks = self._fdesc1_info[op['name']]
        ops.adam_filter_prep(self._fdesc1[off:], 1, blocks, self._small, 0, self.params, self.grads, m, v, alpha, beta1, beta2, eps, gscale, self.dtype)

    def adam_update_small(self, m, v, alpha, beta1, beta2, eps, gscale):
        """The parameters that are not filters (gamma, beta, biases) after every layer went through adam_update_layer."""
        self._filter_descs()
        ops.adam_filter_prep(self._fdesc, 0, 0, self._small, self._n_small, self.params, self.grads, m, v, alpha, beta1, beta2, eps, gscale, self.dtype)
        self._filters_dirty = False

    def _folds(self, op):
        """Inference-only BN folding applies to batch-normalised layers whose output is a plain activation tensor: not the
        image layer (its direct kernel has no bias path) and not a layer fused with its max pool."""
        return (not self.training and self.fold_bn and op['bn'] and op['out'] not in self.fused_pool
                and op['x'] not in self.graph.inputs.values())

    def set_images(self, images, mode=0):
        """images: f32 [B,H,W,3] device tensor (0..255 for mode 0/1).  mode 0 = per_image_standardization
        (train.py:103 / detect.py `std`), 1 = /255 (detect.py `darknet`), 2 = already preprocessed."""
        (inp,) = self.graph.inputs.values()
        assert images.dtype == torch.float32 and images.is_cuda and images.numel() == self.B * inp.h * inp.w * 3
        self.img = images.contiguous()
        ops.image_prep(self.img, self.act[inp][0], self.ws, self.B, inp.h * inp.w, mode)

    def _plan_grad_zeroing(self):
        """Arena ranges that must be zero before backward: filter gradients that accumulate with atomics (several
        pixel ranges per tile).  Layers whose filter gradient is a single range store directly -- at batch 16 those are
        the three 13x13 layers that hold 70 % of the parameters -- and BN / bias gradients are stored by their
        finalisation kernels, so most of the 268 MB arena is never cleared."""
        if False:        # (debugging aid: clear the whole gradient arena every step)
            return [(0, self.n_params)]
        ranges = []
        for op in self.graph.ops:
            if op['kind'] != 'conv':
                continue
            x, out = op['x'], op['out']
            ldx, ldy = self.act[x][1], pad8(op['cout'])
            if ops.conv2d_wgrad_accumulates(self.B, x.h, x.w, op['cin'], ldx, op['cout'], ldy, op['ksize'], self.dtype):
                off, size = self.param_offsets[op['weights'].name]
                ranges.append((off, off + (size + 3) // 4 * 4))
        ranges.sort()
        merged = []
        for a, b in ranges:
            if merged and a <= merged[-1][1]:
                merged[-1] = (merged[-1][0], max(merged[-1][1], b))
            else:
                merged.append((a, b))
        # bridging small gaps costs less than another launch
        out = []
        for a, b in merged:
            if out and a - out[-1][1] <= (1 << 20):
                out[-1] = (out[-1][0], b)
            else:
                out.append((a, b))
        return out

    def zero_grads(self):
        if self._zero_ranges is None:
            self._zero_ranges = self._cur['zero_ranges'] = self._plan_grad_zeroing()
        if self._zero_ranges:
            ops.zero_ranges(self.grads, self._zero_ranges)

    # ---------------------------------------------------------------- forward
    def forward(self):
        self._prepare_filters()
        B = self.B
        if self.training:
            self.state_snap.copy_(self.state)        # this step's shifts (see _alloc_variables)
        pooled = set()               # pool ops already produced by their producer's BN pass in this sweep
        for op in self.graph.ops:
            kind = op['kind']
            if kind == 'conv':
                x, out = op['x'], op['out']
                xb, ldx = self.act[x]
                st = self.conv[op['name']]
                M = B * out.h * out.w
                if op['bn'] and 'fold_bias' in st:
                    ob, ldo = self.act[out]
                    ops.conv2d_bias_leaky(xb, st['Ffwd'], st['fold_bias'], ob, self.conv_ws, B, x.h, x.w, pad8(x.c), ldx, op['cout'], ldo, op['ksize'], LEAKY_ALPHA)
                elif op['bn'] and self._first_fused(op):
                    # image layer: its raw output (the largest tensor of the network) is never stored; the statistics pass and the
                    # BN + leaky + pool pass each recompute it from the image (csrc/conv_first.hip: conv_first_pool_kernel)
                    gamma, beta = self.var[op['gamma'].name], self.var[op['beta'].name]
                    mmean, mvar = self.var[op['moving_mean'].name], self.var[op['moving_variance'].name]
                    pool = self.fused_pool[out]
                    pb, ldp = self.act[pool['out']]
                    if self.training:
                        part = self.parts.acquire(2 * 256 * 32)
                        shift = self.svar[op['moving_mean'].name]       # this step's snapshot, like every other layer: never aliases the updated moving mean
                        ops.first_layer_stats(xb, st['Ffwd'], B, x.h, x.w, shift, self.parts.bufs[part])
                        ops.bn_finalize(self.parts.bufs[part], shift, M, 32, st['mean'], st['var'], mmean, mvar, BN_DECAY)
                        self.parts.consumed(part, cleared=True)
                        mean, var = st['mean'], st['var']
                    else:
                        mean, var = mmean, mvar
                    ops.first_layer_bn_leaky_pool(xb, st['Ffwd'], mean, var, gamma, beta, pb, st.get('pool_idx') if self.training else None,
                                                  B, x.h, x.w, ldp, BN_EPS, LEAKY_ALPHA)
                elif op['bn']:
                    yb, ldy = self.act[op['y']]
                    gamma, beta = self.var[op['gamma'].name], self.var[op['beta'].name]
                    mmean, mvar = self.var[op['moving_mean'].name], self.var[op['moving_variance'].name]
                    fused = self.training and self.fuse_bn_stats and ldy == op['cout']
                    shift = self.svar[op['moving_mean'].name]       # the moving mean as of the start of this step (read-only during the step)
                    produced = self._conv(xb, st['Ffwd'], None, yb, x.h, x.w, pad8(x.c), ldx, op['cout'], ldy, op['ksize'], op['ksize'] ** 2 * op['cin'],
                                          bn_shift=shift if fused else None)
                    fin = None
                    if fused:
                        # batch moments from the partial sums the convolution left behind (shift = the moving mean)
                        part, rows = produced
                        if self.fold_finalize and ops.bn_fin_supported(rows, op['cout'], self.dtype):
                            fin = (part, rows)              # ... summed by the BN-apply kernel below, in its prologue
                        else:
                            if self.sync_bn and self.bn_world > 1:
                                self._sync_bn_sums(self.parts.bufs[part], op['cout'])
                            ops.bn_finalize(self.parts.bufs[part], shift, M * (self.bn_world if self.sync_bn else 1), op['cout'], st['mean'], st['var'],
                                            mmean, mvar, BN_DECAY)
                            self.parts.consumed(part, cleared=True)
                        mean, var = st['mean'], st['var']
                    elif self.training:
                        assert not (self.sync_bn and self.bn_world > 1), 'sync_bn needs the statistics from the convolution epilogue (YOLO2_FUSE_BN_STATS=1, dense output)'
                        ops.bn_stats_ema(yb, st['mean'], st['var'], mmean, mvar, BN_DECAY, self.ws, M, op['cout'])
                        mean, var = st['mean'], st['var']
                    else:
                        mean, var = mmean, mvar
                    pool = self.fused_pool.get(out)
                    if fin is not None:
                        part, rows = fin
                        zbuf, zn = self.parts.take_to_zero(part)
                        if pool is not None:
                            pb, ldp = self.act[pool['out']]
                            ops.bn_leaky_pool_fin(yb, self.parts.bufs[part], rows, shift, mean, var, mmean, mvar, BN_DECAY, gamma, beta, pb, st.get('pool_idx'),
                                                  B, out.h, out.w, op['cout'], ldp, BN_EPS, LEAKY_ALPHA, zbuf, zn, ymax=st.get('pool_ymax'))
                            st['ymax_valid'] = 'pool_ymax' in st
                        elif out in self.fwd_pool:        # both resolutions from one pass (the activation has a second reader)
                            pb, ldp = self.act[self.fwd_pool[out]['out']]
                            ops.bn_leaky_pool_fin(yb, self.parts.bufs[part], rows, shift, mean, var, mmean, mvar, BN_DECAY, gamma, beta, pb, None,
                                                  B, out.h, out.w, op['cout'], ldp, BN_EPS, LEAKY_ALPHA, zbuf, zn, a_full=self.act[out][0])
                            pooled.add(self.fwd_pool[out]['name'])
                        else:
                            ob, ldo = self.act[out]
                            ops.bn_leaky_fin(yb, self.parts.bufs[part], rows, shift, mean, var, mmean, mvar, BN_DECAY, gamma, beta, ob, M, op['cout'], ldo,
                                             BN_EPS, LEAKY_ALPHA, zbuf, zn)
                        self.parts.consumed(part)
                    elif pool is not None:
                        pb, ldp = self.act[pool['out']]
                        ops.bn_leaky_pool(yb, mean, var, gamma, beta, pb, st.get('pool_idx'), B, out.h, out.w, op['cout'], ldp, BN_EPS, LEAKY_ALPHA)
                        st['ymax_valid'] = False
                    else:
                        ob, ldo = self.act[out]
                        ops.bn_leaky(yb, mean, var, gamma, beta, ob, M, op['cout'], ldo, BN_EPS, LEAKY_ALPHA)
                elif op['act']:
                    # un-normalised layer of the YOLO (v1) family: conv / fully connected + biases + leaky_relu in one launch
                    ob, ldo = self.act[out]
                    ops.conv2d_bias_leaky(xb, st['Ffwd'], self.var[op['biases'].name], ob, self.conv_ws, B, x.h, x.w, pad8(x.c), ldx, op['cout'], ldo, op['ksize'], LEAKY_ALPHA)
                else:
                    ob, ldo = self.act[out]
                    self._conv(xb, st['Ffwd'], self.var[op['biases'].name], ob, x.h, x.w, pad8(x.c), ldx, op['cout'], ldo, op['ksize'], op['ksize'] ** 2 * op['cin'])
            elif kind == 'flatten':
                pass                         # the same bytes, re-read as one pixel
            elif kind == 'dropout':
                x, out = op['x'], op['out']
                n = B * x.h * x.w * self.act[x][1]
                mask = self._dropout_mask(op, n)
                fixed = self.dropout_masks is not None
                self._dropout_calls += 1
                ops.dropout(self.act[x][0], self.act[out][0], mask, n, op['keep_prob'], 0 if fixed else self._dropout_seed_for(op))
            elif kind == 'pool':
                x, out = op['x'], op['out']
                if x in self.fused_pool or op['name'] in pooled:
                    continue                 # produced by the conv's fused BN + leaky + pool pass
                assert self.act[x][1] == x.c and self.act[out][1] == out.c
                ops.maxpool_fwd(self.act[x][0], self.act[out][0], B, x.h, x.w, x.c, op['stride'])
            elif kind == 'reorg':
                x, out = op['x'], op['out']
                assert self.act[x][1] == x.c
                ops.reorg(self.act[x][0], self.act[out][0], B, x.h, x.w, x.c, self.act[out][1])
            elif kind == 'concat':
                pass
            else:
                raise ValueError(kind)

    # ---------------------------------------------------------------- backward
    def _grad_sink(self, t, written):
        """Where a consumer writes d/dt: the tensor's gradient buffer for the first writer of this
        backward pass, a temporary that is added afterwards for later ones (passthrough fan-out)."""
        gb, ld = self.gact[t]
        if t not in written:
            written.add(t)
            return gb, ld, None
        tmp = self.tmp_grad.get(t)
        if tmp is None:
            n = self.B * t.h * t.w * ld
            root = self._tmp_roots.get(t.name)          # shared between input sizes (a prefix serves the smaller ones)
            if root is None or root.numel() < n:
                root = self._tmp_roots[t.name] = torch.zeros(n, dtype=self.dtype, device=self.device)
            tmp = self.tmp_grad[t] = root
        n = self.B * t.h * t.w * ld
        return tmp, ld, (lambda: ops.add_inplace(gb, tmp, n))

    def backward(self, on_layer_done=None):
        """Reverse sweep from the gradient already stored for the graph output (written by loss()).
        Per convolution: BN/leaky backward -> dY, then the filter gradient on the side stream overlapped with the data
        gradient (and everything after it) on the main stream: at batch 16 the 13x13 / 26x26 stages launch grids
        that cannot fill the chip on their own.  ``on_layer_done(op, event)`` is called once a layer's parameter
        gradients are enqueued; ``event`` (or None) completes when they are final."""
        B = self.B
        written = set()
        reduced = set()
        inputs = set(self.graph.inputs.values())
        self._phase = 'dgrad'
        if self._has_l2:
            self.reg_loss.zero_()
        main = torch.cuda.current_stream()
        side = self.side_stream if self.overlap_wgrad else None
        slot = 0
        for op in reversed(self.graph.ops):
            kind = op['kind']
            if kind == 'conv':
                x, out = op['x'], op['out']
                st = self.conv[op['name']]
                M = B * out.h * out.w
                cout, k = op['cout'], op['ksize']
                ldy = pad8(cout)
                wgrad_done = False
                gob, ldgo = self.gact[out]
                xb, ldx = self.act[x]
                if op['bn']:
                    yb, _ = self.act[op['y']]
                    gamma, beta = self.var[op['gamma'].name], self.var[op['beta'].name]
                    dgam, dbet = self.gvar[op['gamma'].name], self.gvar[op['beta'].name]
                    pool = self.fused_pool.get(out)
                    # ``pfin`` = (partial rows [2][rows][cout], rows, plane stride): dgamma / dbeta are still unsummed and the apply pass
                    # below sums them in its prologue (one launch instead of reduce-finalise + apply)
                    pfin = None
                    limit = self._fin_limit(cout) if self.fold_finalize else 0
                    first = self._first_fused(op)
                    if first:                 # image layer: y is recomputed from the image inside both backward kernels
                        dpb, lddp = self.gact[pool['out']]
                        part = self.parts.acquire(2 * 256 * 32)
                        ops.first_layer_pool_bwd_reduce(xb, st['Ffwd'], dpb, lddp, st['pool_idx'], st['mean'], st['var'], gamma, beta, self.parts.bufs[part],
                                                        B, x.h, x.w, BN_EPS, LEAKY_ALPHA)
                        ops.bn_part_to_grads(self.parts.bufs[part], 32, dgam, dbet)
                        self.parts.consumed(part, cleared=True)
                    elif pool is not None:      # gradient arrives at the POOLED resolution; routed through the stored arg-max
                        dpb, lddp = self.gact[pool['out']]
                        if limit >= 64 and st.get('ymax_valid'):
                            # only arg-max positions carry gradient: the sums over (dP, y at the arg-max) ARE the layer's sums
                            rows = ops.bn_leaky_bwd_reduce_part(dpb, lddp, st['pool_ymax'], st['mean'], st['var'], gamma, beta, self.ws, limit,
                                                                M // 4, cout, BN_EPS, LEAKY_ALPHA)
                            pfin = (self.ws, rows, rows * cout)
                        elif limit >= 128:
                            rows = ops.bn_leaky_pool_bwd_reduce_part(dpb, lddp, st['pool_idx'], yb, st['mean'], st['var'], gamma, beta, self.ws, limit,
                                                                     B, out.h, out.w, cout, BN_EPS, LEAKY_ALPHA)
                            pfin = (self.ws, rows, rows * cout)
                        else:
                            ops.bn_leaky_pool_bwd_reduce(dpb, lddp, st['pool_idx'], yb, st['mean'], st['var'], gamma, beta, dgam, dbet, self.ws,
                                                         B, out.h, out.w, cout, BN_EPS, LEAKY_ALPHA)
                    elif op['name'] in reduced:
                        pend = self._bz_pending.pop(op['name'], None)      # sums from the consumer's data-gradient epilogue
                        if pend is not None:
                            pfin = (self.parts.bufs[pend[0]], pend[1], 256 * cout, pend[0])
                    elif limit >= 64:
                        rows = ops.bn_leaky_bwd_reduce_part(gob, ldgo, yb, st['mean'], st['var'], gamma, beta, self.ws, limit, M, cout, BN_EPS, LEAKY_ALPHA)
                        pfin = (self.ws, rows, rows * cout)
                    else:
                        ops.bn_leaky_bwd_reduce(gob, ldgo, yb, st['mean'], st['var'], gamma, beta, dgam, dbet, self.ws, M, cout, BN_EPS, LEAKY_ALPHA)
                    slot = (slot + 1) % 3
                    dy = self.dy_ring[slot]
                    if self.dy_free[slot] is not None:
                        main.wait_event(self.dy_free[slot])       # the filter gradient that last read this buffer is done
                    if first:
                        ops.first_layer_pool_bwd_apply(xb, st['Ffwd'], dpb, lddp, st['pool_idx'], st['mean'], st['var'], gamma, beta, dgam, dbet, dy,
                                                       B, x.h, x.w, BN_EPS, LEAKY_ALPHA)
                    elif pfin is not None:
                        own = pfin[3] if len(pfin) > 3 else -1
                        zbuf, zn = self.parts.take_to_zero(own)
                        if pool is not None and self._first_wgrad_fused(op, lddp):
                            # image layer, stored-output path: the BN / leaky / pool backward apply runs inside the filter gradient (its 177 MB output
                            # gradient had one reader; csrc/conv_first.hip conv_first_wgrad_bn_kernel)
                            ops.first_layer_wgrad_bn(xb, yb, dpb, lddp, st['pool_idx'], st['mean'], st['var'], gamma, beta, pfin[0], pfin[1], pfin[2],
                                                     dgam, dbet, self.gvar[op['weights'].name], B, x.h, x.w, op['cin'], BN_EPS, LEAKY_ALPHA, zbuf, zn)
                            self._l2(op)
                            wgrad_done = True
                        elif pool is not None:
                            ops.bn_leaky_pool_bwd_apply_fin(dpb, lddp, st['pool_idx'], yb, st['mean'], st['var'], gamma, beta, pfin[0], pfin[1], pfin[2],
                                                            dgam, dbet, dy, B, out.h, out.w, cout, BN_EPS, LEAKY_ALPHA, zbuf, zn)
                        else:
                            ops.bn_leaky_bwd_apply_fin(gob, ldgo, yb, st['mean'], st['var'], gamma, beta, pfin[0], pfin[1], pfin[2], dgam, dbet, dy,
                                                       M, cout, BN_EPS, LEAKY_ALPHA, zbuf, zn)
                        if own >= 0:
                            self.parts.consumed(own)
                    elif pool is not None:
                        ag, ab = self._sync_bn_grads(dgam, dbet, cout)
                        ops.bn_leaky_pool_bwd_apply(dpb, lddp, st['pool_idx'], yb, st['mean'], st['var'], gamma, beta, ag, ab, dy,
                                                    B, out.h, out.w, cout, BN_EPS, LEAKY_ALPHA)
                    else:
                        ag, ab = self._sync_bn_grads(dgam, dbet, cout)
                        ops.bn_leaky_bwd_apply(gob, ldgo, yb, st['mean'], st['var'], gamma, beta, ag, ab, dy, M, cout, BN_EPS, LEAKY_ALPHA)
                    ring = True
                elif op['act']:
                    # un-normalised layer + leaky_relu (YOLO v1): dZ from the layer's output sign, then the biased-layer path
                    assert ldgo == ldy and self.act[out][1] == ldy
                    slot = (slot + 1) % 3
                    dy = self.dy_ring[slot]
                    if self.dy_free[slot] is not None:
                        main.wait_event(self.dy_free[slot])
                    ops.leaky_bwd(self.act[out][0], gob, dy, M * ldy, LEAKY_ALPHA)
                    ops.bias_grad(dy, ldy, self.gvar[op['biases'].name], self.ws, M, cout)
                    ring = True
                else:
                    dy = gob
                    assert ldgo == ldy
                    ops.bias_grad(dy, ldy, self.gvar[op['biases'].name], self.ws, M, cout)
                    ring = False
                done = None
                if wgrad_done:
                    pass
                elif side is not None and M <= self.overlap_max_m:
                    ready = torch.cuda.Event()
                    ready.record(main)
                    side.wait_event(ready)
                    with torch.cuda.stream(side):
                        ops.conv2d_wgrad(xb, dy, self.gvar[op['weights'].name], B, x.h, x.w, op['cin'], ldx, cout, ldy, k)
                        self._l2(op)
                        done = torch.cuda.Event()
                        done.record(side)
                    if ring:
                        self.dy_free[slot] = done
                else:
                    ops.conv2d_wgrad(xb, dy, self.gvar[op['weights'].name], B, x.h, x.w, op['cin'], ldx, cout, ldy, k)
                    self._l2(op)
                if x not in inputs:
                    prod = self.bn_bwd_fused.get(op['name'])
                    dst, ldd, fin = self._grad_sink(x, written)
                    if prod is not None and fin is None:
                        pst = self.conv[prod['name']]
                        pend = self._conv(dy, st['Fdgr'], None, dst, x.h, x.w, ldy, ldy, op['cin'], ldd, k, k * k * cout,
                                          bn_bwd=(self.act[prod['y']][0], pst['mean'], pst['var'], self.var[prod['gamma'].name], self.var[prod['beta'].name],
                                                  self.gvar[prod['gamma'].name], self.gvar[prod['beta'].name]))
                        reduced.add(prod['name'])
                        if pend is not None:
                            self._bz_pending[prod['name']] = pend
                    else:
                        self._conv(dy, st['Fdgr'], None, dst, x.h, x.w, ldy, ldy, op['cin'], ldd, k, k * k * cout)
                    if fin:
                        fin()
                if on_layer_done is not None:
                    on_layer_done(op, done)
            elif kind == 'pool':
                x, out = op['x'], op['out']
                if x in self.fused_pool:
                    continue                 # routed inside the producer's fused BN backward
                if x in written and op['stride'] == 2 and self.gact[x][1] == x.c:
                    # second writer of this gradient (the passthrough branch wrote first): accumulate in place, one launch
                    ops.maxpool_bwd_acc(self.act[x][0], self.gact[out][0], self.gact[x][0], B, x.h, x.w, x.c)
                    continue
                dst, ldd, fin = self._grad_sink(x, written)
                assert ldd == x.c
                ops.maxpool_bwd(self.act[x][0], self.gact[out][0], dst, B, x.h, x.w, x.c, op['stride'])
                if fin:
                    fin()
            elif kind == 'reorg':
                x, out = op['x'], op['out']
                dst, ldd, fin = self._grad_sink(x, written)
                assert ldd == x.c
                ops.reorg_bwd(self.gact[out][0], self.gact[out][1], dst, B, x.h, x.w, x.c)
                if fin:
                    fin()
            elif kind == 'concat':
                for v in op['inputs']:
                    written.add(v)           # their gradients are slices of the concat gradient
            elif kind == 'flatten':
                written.add(op['x'])         # the gradient of the flat view IS the gradient of the tensor
            elif kind == 'dropout':
                x, out = op['x'], op['out']
                dst, ldd, fin = self._grad_sink(x, written)
                n = B * x.h * x.w * ldd
                ops.dropout_bwd(self.gact[out][0], self._dropout_mask(op, n), dst, n, op['keep_prob'])
                if fin:
                    fin()
        self._phase = 'fwd'
        if side is not None:
            main.wait_stream(side)           # every filter gradient is final before the optimizer / the caller reads them

    def _sync_bn_sums(self, part, C):
        """Forward, sync_bn: the partial rows [2][256][C] of THIS replica become the sums over ALL replicas (in row 0; bn_finalize then sees
        world x M samples).  Every replica holds the same shift (the moving mean), so the shifted sums simply add."""
        import torch.distributed as dist
        sums = self._bn_sync_buf(2 * C)
        ops.bn_part_to_grads(part, C, sums[:C], sums[C:])                # column sums of both planes; clears the rows
        dist.all_reduce(sums, group=self.bn_group)
        part[:C].copy_(sums[:C])
        part[BN_PART_ROWS * C:BN_PART_ROWS * C + C].copy_(sums[C:])      # (plane 1 starts BN_PART_ROWS rows after plane 0)

    def _sync_bn_grads(self, dgam, dbet, C):
        """Backward, sync_bn: the apply pass needs sum(dz * xhat), sum(dz) over ALL replicas' pixels (divided by the global count: passed
        pre-divided by the world size, the kernel divides by this replica's pixel count).  dgamma / dbeta themselves stay this replica's
        sums -- the gradient exchange averages them like every other parameter gradient."""
        if not (self.sync_bn and self.bn_world > 1):
            return dgam, dbet
        import torch.distributed as dist
        sums = self._bn_sync_buf(2 * C)
        sums[:C].copy_(dgam)
        sums[C:].copy_(dbet)
        dist.all_reduce(sums, group=self.bn_group)
        sums.mul_(1.0 / self.bn_world)
        return sums[:C], sums[C:]

    def _bn_sync_buf(self, n):
        buf = getattr(self, '_bn_sync', None)
        if buf is None or buf.numel() < n:
            buf = self._bn_sync = torch.zeros(max(n, 4096), dtype=torch.float32, device=self.device)
        return buf[:n]

    def _first_fused(self, op):
        """The image layer (3 channels in an 8-wide pixel, 32 filters, 3x3, batch-normalised, followed only by a 2x2 pool) takes the
        recompute-instead-of-store kernels; YOLO2_FUSE_FIRST=0 keeps the stored-output path (A/B, and tests that read the raw output)."""
        if self.fuse_first == '0' or (self.training and self.fuse_first != '1'):
            return False
        x, out = op['x'], op['out']
        return (op['bn'] and x in self.graph.inputs.values() and op['ksize'] == 3 and op['cin'] == 3 and self.act[x][1] == 8 and op['cout'] == 32
                and out in self.fused_pool and self.act[self.fused_pool[out]['out']][1] >= 32 and x.h % 2 == 0 and x.w % 2 == 0
                and (not self.training or self.fuse_bn_stats))

    def _first_wgrad_fused(self, op, lddp):
        """Image layer on the stored-output path: filter gradient with the BN / leaky / pool backward apply inside (YOLO2_FUSE_FIRST_WGRAD=0: the two
        launches, A/B)."""
        x, out = op['x'], op['out']
        return (self.fuse_first_wgrad and op['bn'] and x in self.graph.inputs.values() and op['ksize'] == 3 and op['cin'] <= 8 and self.act[x][1] == 8
                and op['cout'] == 32 and out in self.fused_pool and lddp >= 32 and lddp % 8 == 0 and x.h % 2 == 0 and x.w % 2 == 0
                and 'yolo2_first_layer_wgrad_bn' not in _lib.MISSING)

    def _fin_limit(self, C):
        """Most partial rows a reduce_part launch may leave for a *_fin consumer with C channels (0: the shape does not qualify)."""
        v = self._fin_rows_limit.get(C)
        if v is None:
            v = self._fin_rows_limit[C] = ops.bn_fin_rows_limit(C, self.dtype)
        return v

    def _l2(self, op):
        """slim.l2_regularizer on a layer's weights (YOLO v1 fully connected layers): gradient and loss term, on the stream of the
        filter gradient it follows."""
        if op.get('l2', 0.0) > 0.0:
            w = self.var[op['weights'].name]
            ops.l2_regularizer(w, self.gvar[op['weights'].name], w.numel(), op['l2'], self.reg_loss)

    def _dropout_seed_for(self, op):
        """Seed of one dropout draw: a hash of (engine seed, data-parallel rank, global step, dropout layer) -- independent masks per
        replica like the reference's per-tower draws, and no replay of the sequence after a restart (never 0: 0 selects fixed masks)."""
        idx = [o['name'] for o in self.graph.ops if o['kind'] == 'dropout'].index(op['name'])
        h = (self.dropout_seed * 0x9E3779B97F4A7C15 + self.dropout_rank * 0xC2B2AE3D27D4EB4F + self.dropout_step * 0x165667B19E3779F9 + idx * 0xD6E8FEB86659FD93)
        h &= (1 << 64) - 1
        h ^= h >> 31
        return (h % ((1 << 62) - 1)) + 1

    def _dropout_mask(self, op, n):
        if self.dropout_masks is not None:
            return self.dropout_masks[op['name']]
        m = self._masks.get(op['name'])
        if m is None or m.numel() < n:
            m = self._masks[op['name']] = torch.zeros(n, dtype=torch.uint8, device=self.device)
        return m

    def output(self):
        return self.graph.ops[-1]['out']

"""Training / detection sessions: the counterpart of what the reference's callers drive through
``slim.learning.train`` (train.py:109-145) and ``sess.run`` (detect.py:69-71)."""
import math
import os

import numpy as np
import torch

from . import ops
from .engine import Engine, layer_end_offsets
from .graph import pad8
from .model.yolo2 import OBJECTIVE_KEYS
from .optim import Optimizer, learning_rate_fn
from .parallel import GradReducer


def _label_shapes(B, cells, C):
    return [(B, cells), (B, cells, C), (B, cells, 4), (B, cells, 2), (B, cells, 2), (B, cells)]


class TrainSession(object):
    """One training replica.  ``step`` = per-image standardisation -> forward (batch-stat BN, EMA
    update first: [TF-sem] UPDATE_OPS) -> loss + its gradient -> backward -> (all-reduce) -> optional
    per-tensor clip -> optimizer; all asynchronous on the current stream."""

    def __init__(self, builder, batch_size, dtype='bf16', optimizer='adam', learning_rate=1e-6, gradient_clip=0.0,
                 config=None, seed=0, world_size=1, bucket_mb=64.0, preprocess_mode=0, sizes=None, grad_dtype='f32', comm_cus=None, comm_timing=False,
                 sync_bn=False, shard_optimizer=False):
        """``sizes``: optional list of (width, height) input sizes for multi-scale training (BASELINE configs[3]); buffers are
        allocated once for the largest, ``set_size`` switches between them, the builder's configured size is selected first."""
        assert builder.training, 'call builder(data, training=True) first'
        self.builder = builder
        self.B = batch_size
        own = (builder.width, builder.height)
        traced = {own: (builder.graph, builder.model)}
        self.v1 = getattr(builder, 'family', 'yolo2') == 'yolo'       # YOLO (v1): linear fully connected head, boxes_per_cell instead of anchors
        assert not (self.v1 and sizes), 'the fully connected YOLO (v1) head fixes the input size'
        for wh in (sizes or []):
            wh = (int(wh[0]), int(wh[1]))
            if wh not in traced:
                traced[wh] = builder.trace(wh[0], wh[1], training=True)
        largest = max(traced, key=lambda wh: wh[0] * wh[1])
        assert all(w <= largest[0] and h <= largest[1] for w, h in traced), 'one size must contain all the others'
        self.engine = Engine(traced[largest][0], batch_size, dtype, training=True, seed=seed, sync_bn=sync_bn and world_size > 1,
                             side_priority=-1 if world_size == 1 else 0)
        e = self.engine
        if e.sync_bn:                # batch moments and BN-backward sums over all replicas ([mi355x] sync_bn; default: replica-local like N reference processes)
            import torch.distributed as dist
            # a group of its own: a process group's collectives run in order on one internal stream, so the small per-layer BN exchanges
            # would otherwise queue behind whatever 64 MB gradient bucket is on the wire (every rank builds its session: new_group is collective)
            e.bn_group, e.bn_world = (dist.new_group() if dist.is_initialized() else dist.group.WORLD), int(world_size)
        self.models = {wh: gm[1] for wh, gm in traced.items()}
        m0 = traced[largest][1]
        self.A, self.C = (m0.boxes_per_cell if self.v1 else len(m0.anchors)), m0.classes
        dev = e.device
        self.anchors = None if self.v1 else torch.from_numpy(m0.anchors.reshape(-1)).to(dev)
        self._labels = {wh: [torch.zeros(*s, dtype=torch.float32, device=dev) for s in _label_shapes(batch_size, mdl.cells, self.C)]
                        for wh, mdl in self.models.items()}
        self.objectives_dev = torch.zeros(4, dtype=torch.float32, device=dev)
        self.loss_ws = torch.zeros(ops.loss_ws_floats(batch_size, m0.cells, self.A), dtype=torch.float32, device=dev)
        for wh, (g, _) in traced.items():
            if wh != largest:
                e.add_size(g)
        self.size = None
        self.set_size(*own)
        self.hparam = [builder.hparam[k] for k in OBJECTIVE_KEYS]
        self.optimizer = Optimizer(optimizer, config if config is not None else builder.config, e.n_params, dev)
        self.lr_fn = learning_rate_fn(config if config is not None else builder.config, learning_rate)
        self.gradient_clip = float(gradient_clip)
        self.clip_ws = torch.zeros(ops.workspace_bytes('clip', e.n_seg) // 8, dtype=torch.float64, device=dev)
        self.global_step = 0
        self.world_size = world_size
        self.preprocess_mode = preprocess_mode
        # [mi355x] shard_optimizer: reduce-scatter / update 1/world / all-gather instead of all-reduce + the replicated optimizer pass
        # (parallel.GradReducer); per-tensor clipping needs every complete gradient first, so it keeps the replicated form
        self.shard_optimizer = bool(shard_optimizer) and world_size > 1 and self.gradient_clip <= 0
        if shard_optimizer and world_size > 1 and not self.shard_optimizer:
            import logging
            logging.getLogger(__name__).warning('[mi355x] shard_optimizer is ignored: gradient_clip = %g needs every complete gradient, the update stays replicated',
                                                self.gradient_clip)
        # False between a sharded update and gather_optimizer_state(): this rank's optimizer slots then hold other ranks' shards from an
        # earlier gather (or their initial values) -- checkpoint.save / tf_checkpoint.save refuse to write them
        self.optimizer_state_complete = True
        self.reducer = GradReducer(e.grads, list(e.param_offsets.values()), bucket_mb, grad_dtype=grad_dtype, timing=comm_timing,
                                   shard_params=e.params if self.shard_optimizer else None) if world_size > 1 else None
        if world_size > 1:
            # the collectives' persistent kernels hold CUs for their whole duration: stream-K launches (one workgroup per CU) are sized
            # for what is left ([mi355x] comm_cus, default 32 -- an RCCL ring's channel count on this part)
            cfg = config if config is not None else builder.config
            reserve = comm_cus if comm_cus is not None else (cfg.getint('mi355x', 'comm_cus') if cfg is not None and cfg.has_option('mi355x', 'comm_cus') else 32)
            total = torch.cuda.get_device_properties(dev).multi_processor_count
            ops.set_stream_workgroups(max(64, total - int(reserve)) if reserve > 0 else 0)
        if world_size > 1 and torch.distributed.is_available() and torch.distributed.is_initialized():
            e.dropout_rank = torch.distributed.get_rank()
        self.bucketed_update = True
        self.fuse_adam_prep = os.environ.get('YOLO2_FUSE_ADAM_PREP', '1') != '0'      # A/B: 0 = Adam launch + separate operand re-layout at the next forward
        # Candidate, OFF by default (YOLO2_EARLY_ADAM=1 enables it): single process, Adam, no clipping -- a layer's filter is updated as soon as
        # its gradient is final, on a third stream, while backward goes on (the update is HBM-bound, 0.36-0.41 ms as one launch at the end of the
        # step; the 13x13 layers whose gradients come first hold 84 % of the parameters).  Bit-identical to the one-launch form (same kernel per
        # layer, tests/test_network_gpu.py).  Measured SLOWER: 3.72 vs 3.62 ms per step (profiles/r05_early_adam.txt) -- the update's thousands
        # of small workgroups (16.6 KB of LDS each) occupy CUs that the 150 KB convolution workgroups then cannot be placed on.
        self.early_adam = (os.environ.get('YOLO2_EARLY_ADAM', '0') == '1' and world_size == 1 and optimizer == 'adam' and self.fuse_adam_prep
                           and self.gradient_clip <= 0 and not getattr(e, '_has_l2', False) and dev.type == 'cuda')
        self.opt_stream = torch.cuda.Stream(device=dev) if self.early_adam else None
        self._early_pending = False
        self._in_step = False            # the early-update candidate only runs inside step(): forward_backward() + apply_gradients() as one unit
        # device-detected failures (a stream-K tile owner that gave up, conv_shared.h) polled every step without a synchronisation
        self.async_errors = ops.AsyncErrorPoll() if dev.type == 'cuda' else None
        # arena offset below which every gradient is final once a given layer's backward has run
        self._layer_end = layer_end_offsets(e.graph, e.param_offsets)

    def set_size(self, width, height):
        """Input size of the following steps (one of the sizes the session was built with); weights, statistics, optimizer state
        and global_step carry over.  Labels must be uploaded for the new grid."""
        wh = (int(width), int(height))
        self.engine.set_size(wh[1], wh[0])
        self.model = self.models[wh]
        self.labels = self._labels[wh]
        self.size = wh

    def upload_labels(self, labels):
        for dst, src in zip(self.labels, labels):
            dst.copy_(torch.from_numpy(np.ascontiguousarray(src, np.float32).reshape(dst.shape)), non_blocking=True)

    def forward_backward(self, images, defer_collectives=False):
        """Gradients are final (all-reduced) on return unless ``defer_collectives``: then the buckets may still be on the wire
        and ``apply_gradients`` consumes them one by one (what ``step`` does)."""
        e, m = self.engine, self.model
        assert not self._early_pending, 'a layer-wise early update is half applied: apply_gradients() must follow the forward_backward() of step()'
        e.dropout_step = self.global_step
        e.zero_grads()
        e.set_images(images, self.preprocess_mode)
        e.forward()
        out = e.output()
        logits, ld = e.act[out]
        dlogits, _ = e.gact[out]
        if self.v1:
            ops.yolo1_loss(logits, ld, self.labels, self.hparam, self.objectives_dev, dlogits, self.loss_ws, self.B, m.cell_height, m.cell_width, self.A, self.C)
        else:
            # (the four objective values are reduced from the partial sums when fetch() asks for them)
            ops.loss_partials(logits, ld, self.anchors, self.labels, self.hparam, dlogits, self.loss_ws, self.B, m.cell_height, m.cell_width, self.A, self.C)
            self._objectives_pending = (m.cell_height, m.cell_width)
        if self.reducer is not None:
            self.reducer.begin()
            # optimizer sharding only inside step(): a bare forward_backward() promises complete (all-reduced) gradients and no update
            self.reducer.shard = self.shard_optimizer and defer_collectives
            self._sharded_pending = self.reducer.shard
            if self.reducer.shard:
                # the update of a bucket's shard runs inside the bucket's chain on the communication stream, while backward continues
                lr, t, gs = self.lr_fn(self.global_step), self.global_step + 1, 1.0 / self.world_size
                self.reducer.update_fn = lambda lo, hi: self.optimizer.apply(e.params, e.grads, lr, t, gs, lo, hi)
            e.backward(on_layer_done=lambda op, ev: self.reducer.ready_upto(self._layer_end[op['name']], ev))
            # without clipping the optimizer consumes the buckets as they arrive (apply_gradients); clipping needs them all
            self._deferred = defer_collectives and self.gradient_clip <= 0 and self.bucketed_update
            self.reducer.finish(wait=not self._deferred)
        elif self.early_adam and defer_collectives and self._in_step:
            hp = self.optimizer.hp
            t = self.global_step + 1
            alpha = self.lr_fn(self.global_step) * math.sqrt(1.0 - hp['beta2'] ** t) / (1.0 - hp['beta1'] ** t)
            main, opt, m_, v_ = torch.cuda.current_stream(), self.opt_stream, self.optimizer.slots[0], self.optimizer.slots[1]

            def update_layer(op, wgrad_done):
                # behind the layer's data gradient on the main stream (it reads the operand the update rewrites) and its filter gradient
                ev = torch.cuda.Event()
                ev.record(main)
                opt.wait_event(ev)
                if wgrad_done is not None:
                    opt.wait_event(wgrad_done)
                with torch.cuda.stream(opt):
                    e.adam_update_layer(op, m_, v_, alpha, hp['beta1'], hp['beta2'], hp['epsilon'], 1.0)
            e.backward(on_layer_done=update_layer)
            self._early_pending = (alpha, hp['beta1'], hp['beta2'], hp['epsilon'])
        else:
            e.backward()

    def apply_gradients(self):
        e = self.engine
        if self.gradient_clip > 0:
            # [TF-sem] clip happens on the (averaged) gradient: scale first when data-parallel
            if self.world_size > 1:
                ops.scale(e.grads, e.n_params, 1.0 / self.world_size)
            ops.clip_by_norm(e.grads, e.seg_off, e.n_seg, self.gradient_clip, self.clip_ws)
            gscale = 1.0
        else:
            gscale = 1.0 / self.world_size
        lr = self.lr_fn(self.global_step)
        if self._early_pending:
            # every filter was updated during backward (forward_backward); the small parameters follow once the updates have drained
            alpha, b1, b2, eps = self._early_pending
            self._early_pending = False
            torch.cuda.current_stream().wait_stream(self.opt_stream)
            e.adam_update_small(self.optimizer.slots[0], self.optimizer.slots[1], alpha, b1, b2, eps, 1.0)
            self.global_step += 1
            return
        if self.reducer is not None and getattr(self, '_sharded_pending', False):
            # the buckets' chains (reduce-scatter, shard update, all-gather) were enqueued during backward: wait for the last of them
            self.reducer.finish(wait=True)
            self._deferred = False
            self._sharded_pending = False
            self.optimizer_state_complete = False      # own shards only, until gather_optimizer_state()
            self.global_step += 1
            e._filters_dirty = True
        elif self.reducer is not None and getattr(self, '_deferred', False):
            # data parallel: update bucket k while the all-reduces of buckets k+1.. are still in flight -- only the last
            # (smallest: the early layers) bucket's collective is exposed, and the 1.9 GB optimizer pass hides the rest
            for lo, hi in self.reducer.completed_buckets():
                self.optimizer.apply(e.params, e.grads, lr, self.global_step + 1, gscale, lo, hi)
            self._deferred = False
            self.global_step += 1
            e._filters_dirty = True
        elif self.optimizer.name == 'adam' and self.fuse_adam_prep:
            # Adam + the operand layouts of the updated filters in one launch: the next forward finds its filters prepared
            self.optimizer.apply_fused_with_filter_prep(e, lr, self.global_step + 1, gscale)
            self.global_step += 1
        else:
            self.optimizer.apply(e.params, e.grads, lr, self.global_step + 1, gscale)
            self.global_step += 1
            e._filters_dirty = True

    def gather_optimizer_state(self):
        """With optimizer sharding every rank holds valid optimizer slots for its own shards only: all-gathers them (collective: every rank
        calls it) so that the rank that writes the checkpoint has the complete state.  No-op for the replicated update."""
        if self.reducer is not None and self.shard_optimizer:
            self.reducer.shard = True
            self.reducer.gather_slots(self.optimizer.slots)
        self.optimizer_state_complete = True

    def step(self, images, labels=None):
        # a failure the device reported during an EARLIER step (completed snapshot): a single process raises here, at most eight steps late instead of one
        # summary interval; data-parallel callers read async_error_pending() / device_error() and agree on it first (train.py) -- a rank
        # that raised alone would leave its peers waiting in the next collective
        if self.async_errors is not None and self.world_size == 1:
            self.async_errors.raise_if_pending()
        if labels is not None:
            self.upload_labels(labels)
        self._in_step = True
        try:
            self.forward_backward(images, defer_collectives=True)
            self.apply_gradients()
        finally:
            self._in_step = False
        if self.async_errors is not None and self.global_step % 8 == 0:      # (a 32-byte device-to-host copy in the stream: polled every step it cost 0.5 % of it)
            self.async_errors.snapshot()

    def async_error_pending(self):
        """True when a completed per-step snapshot shows a device-detected failure (no synchronisation)."""
        return self.async_errors is not None and self.async_errors.pending()

    def device_error(self):
        """Synchronises and RETURNS the device-detected failure (an exception object) or None: data-parallel callers put it through agree()
        and raise on every rank (train.py), single-process callers may simply call fetch()."""
        try:
            ops.check_async_errors()
        except RuntimeError as exc:
            return exc
        return None

    def fetch(self):
        """Synchronises and returns {'total_loss', 'iou_best', 'iou_normal', 'coords', 'prob'} of the last step
        (the five scalars the reference summarises, config.ini:63)."""
        if getattr(self, '_objectives_pending', None):
            ops.loss_objectives(self.loss_ws, self.objectives_dev, self.B, self._objectives_pending[0], self._objectives_pending[1], self.A)
            self._objectives_pending = None
        ops.check_async_errors()           # device-detected failures (a stream-K hand-off that gave up) become exceptions here
        vals = self.objectives_dev.cpu().numpy().astype(np.float64)
        out = {k: float(v) for k, v in zip(OBJECTIVE_KEYS, vals)}
        out['regularization'] = float(self.engine.reg_loss.item())        # slim.l2_regularizer terms (YOLO v1 fully connected layers; 0 for yolo2)
        out['total_loss'] = float(sum(v * w for v, w in zip(vals, self.hparam))) + out['regularization']
        self.builder.objectives.update({k: out[k] for k in OBJECTIVE_KEYS})
        return out


class DetectSession(object):
    """Forward with moving-average BN + decode + batched on-GPU NMS (detect.py:69-80)."""

    def __init__(self, builder, batch_size=1, dtype='bf16', seed=0):
        assert not builder.training
        self.builder = builder
        self.model = m = builder.model
        self.engine = Engine(builder.graph, batch_size, dtype, training=False, seed=seed)
        dev = self.engine.device
        self.v1 = getattr(builder, 'family', 'yolo2') == 'yolo'
        self.B, self.A, self.C = batch_size, (m.boxes_per_cell if self.v1 else len(m.anchors)), m.classes
        n = m.cells * self.A
        self.N = n
        self.anchors = None if self.v1 else torch.from_numpy(m.anchors.reshape(-1)).to(dev)
        self.conf = torch.zeros(batch_size, n, self.C, dtype=torch.float32, device=dev)
        self.xy_min = torch.zeros(batch_size, n, 2, dtype=torch.float32, device=dev)
        self.xy_max = torch.zeros(batch_size, n, 2, dtype=torch.float32, device=dev)
        self.order = torch.zeros(batch_size, n, dtype=torch.int32, device=dev)
        self.nan_flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self._attrs = None
        m.bind(self)                 # Model.conf / xy_min / xy_max / iou / prob / xy / wh read this session's buffers
        self.nms_ws = torch.zeros(ops.workspace_bytes('nms', batch_size, n, self.C) // 4, dtype=torch.int32, device=dev)
        self.async_errors = ops.AsyncErrorPoll()

    def run(self, images, preprocess_mode=0, check_numerics=True):
        """images: device f32 [B,H,W,3].  Returns device tensors conf [B,N,C], xy_min, xy_max [B,N,2]
        (cell units).  Raises FloatingPointError on NaN/Inf like tf.check_numerics (detect.py:70)."""
        e, m = self.engine, self.model
        e.set_images(images, preprocess_mode)
        e.forward()
        logits, ld = e.act[e.output()]
        self._attrs_valid = False
        self.nan_flag.zero_()
        if self.v1:
            ops.yolo1_head_decode(logits, ld, self.conf, self.xy_min, self.xy_max, self.nan_flag, self.B, m.cell_height, m.cell_width, self.A, self.C)
        else:
            ops.head_decode(logits, ld, self.anchors, self.conf, self.xy_min, self.xy_max, self.nan_flag, self.B, m.cell_height,
                            m.cell_width, self.A, self.C)
        if check_numerics:
            bad = int(self.nan_flag.item()) != 0           # (synchronises: the device-side failure check below then costs one small copy)
            ops.check_async_errors()                      # a stream-K hand-off that gave up: these scores are garbage -- raise like check_numerics
            if bad:
                raise FloatingPointError('conf/xy_min/xy_max : Tensor had NaN or Inf values')
        else:
            # benchmark / pipelined callers: no synchronisation here; a failure of an EARLIER run raises now, this run's is polled by the next
            self.async_errors.raise_if_pending()
            self.async_errors.snapshot()
        return self.conf, self.xy_min, self.xy_max

    def attrs(self):
        """iou [B,N], prob [B,N,C], xy, wh [B,N,2] of the last run (model/yolo2/__init__.py:36-56), decoded on first use."""
        if self._attrs is None:
            dev = self.engine.device
            self._attrs = {'iou': torch.zeros(self.B, self.N, dtype=torch.float32, device=dev),
                           'prob': torch.zeros(self.B, self.N, self.C, dtype=torch.float32, device=dev),
                           'xy': torch.zeros(self.B, self.N, 2, dtype=torch.float32, device=dev),
                           'wh': torch.zeros(self.B, self.N, 2, dtype=torch.float32, device=dev)}
            self._attrs_valid = False
        if not getattr(self, '_attrs_valid', False):
            e, m, a = self.engine, self.model, self._attrs
            logits, ld = e.act[e.output()]
            ops.head_decode_attrs(logits, ld, self.anchors, a['iou'], a['prob'], a['xy'], a['wh'], self.B, m.cell_height, m.cell_width, self.A, self.C)
            self._attrs_valid = True
        return self._attrs

    def nms(self, threshold=0.3, threshold_iou=0.4):
        """In-place batched NMS on the decoded boxes; returns order [B,N] (reference list order)."""
        ops.nms(self.conf, self.xy_min, self.xy_max, self.order, self.nms_ws, self.B, self.N, self.C, float(threshold), float(threshold_iou))
        return self.order

    def detect(self, images, threshold=0.3, threshold_iou=0.4, preprocess_mode=0):
        self.run(images, preprocess_mode, check_numerics=False)
        self.nms(threshold, threshold_iou)
        return self.conf, self.xy_min, self.xy_max, self.order

"""Data parallelism over one 8xMI355X node: one process per GPU, full replica per GPU, gradients
summed with RCCL (torch.distributed backend "nccl") over xGMI.  New work relative to the reference,
which has no collective at all (README.md:99 "Multi-GPU supporting" unchecked).

The flat gradient arena is laid out in reverse creation order, so as backward proceeds the finished
region grows from offset 0.  ``GradReducer`` cuts it into buckets sized for xGMI (point-to-point
links, ~153 GB/s each: a few large messages beat many small ones) and launches each bucket's
all-reduce on a dedicated communication stream as soon as backward has passed it, so the
collective overlaps the remaining backward kernels; the optimizer waits for the last bucket.  The
1/world averaging is folded into the optimizer kernel (``gscale``).  Batch-norm statistics stay
local to each replica."""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def sync_replicas(session, src=0, always=False):
    """Makes every replica identical to rank ``src``: parameters, BN moving statistics, optimizer slots and global_step
    (one broadcast each over the flat arenas).  Called once after rank 0 has restored / transferred a checkpoint, so ranks
    can never start from different files (or one of them from a logdir another rank is deleting)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not always):
        return
    e = session.engine
    for t in [e.params, e.state] + list(session.optimizer.slots):
        dist.broadcast(t, src=src)
    step = torch.tensor([session.global_step], dtype=torch.int64, device=e.params.device)
    dist.broadcast(step, src=src)
    session.global_step = int(step.item())
    e._filters_dirty = True


def agree(flags, device):
    """Element-wise MAX over ranks of a short list of integer flags (identity at world size 1): lets every rank take the same
    branch on a decision only one of them can make (a non-finite loss on its shard, rank 0's wall clock)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [int(f) for f in flags]
    t = torch.tensor([int(f) for f in flags], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [int(v) for v in t.tolist()]


def make_buckets(offsets_sizes, total, bucket_elems):
    """Splits [0, total) at variable boundaries into contiguous buckets of >= bucket_elems elements
    (the last one takes the remainder).  ``offsets_sizes``: iterable of (offset, size), any order; a boundary = the start of a variable
    (the arena may pad between variables: engine.ARENA_ALIGN)."""
    bounds = sorted({o for o, _ in offsets_sizes if o > 0} | {total})
    buckets, start = [], 0
    for b in bounds:
        if b - start >= bucket_elems:
            buckets.append((start, b))
            start = b
    if start < total:
        buckets.append((start, total))
    return buckets


def shard_of(s, e, world, rank, align=64):
    """Optimizer sharding of the bucket [s, e): ``world`` equal shards of L = floor((e - s) / (world * align)) * align elements (what
    reduce-scatter / all-gather need) followed by a remainder of fewer than world * align elements that stays replicated.
    -> (own_lo, own_hi, sharded_end): this rank updates [own_lo, own_hi) and the remainder [sharded_end, e)."""
    L = ((e - s) // (world * align)) * align
    return s + rank * L, s + (rank + 1) * L, s + world * L


class GradReducer(object):
    def __init__(self, grads, offsets_sizes, bucket_mb=64.0, group=None, always_reduce=False, grad_dtype='f32', timing=False,
                 shard_params=None, rank=None):
        """``always_reduce``: issue the collectives even in a one-rank group (RCCL smoke tests on a single GPU).
        ``shard_params`` (the flat parameter arena; [mi355x] shard_optimizer): optimizer sharding.  A bucket is then reduce-SCATTERED, this
        rank updates its 1/world shard through ``update_fn(lo, hi)`` (set by the session before backward: the optimizer kernel over a
        slice of the arenas) and the updated parameters are all-gathered -- all three on the communication stream, bucket by bucket while
        backward is still running on earlier layers.  The replicated optimizer pass (1.9 GB of HBM traffic per rank and step, the same on
        every rank) shrinks to 1/world of it; the wire carries the same bytes as the all-reduce it replaces (ring all-reduce = reduce-
        scatter + all-gather).  Master parameters are only read by the next forward's operand preparation, so updating the late layers'
        parameters while the early layers' backward runs is safe; optimizer slots are valid for a rank's own shards only
        (``gather_slots`` before a checkpoint).
        ``grad_dtype`` 'bf16': a bucket is rounded to bf16 into a wire buffer, all-reduced there and widened back into the f32 arena --
        half the bytes per link (134 MB instead of 269 MB per step and rank); the sum over ranks is then formed in bf16 by
        the collective, which costs ~3 significant digits of the SUMMED gradient (Adam normalises its scale away; the default stays f32).
        ``timing``: timed events around every collective and at the point where the consumer starts to wait for it, for
        ``exposed_times()`` (costs nothing when off)."""
        self.grads = grads
        assert grad_dtype in ('f32', 'bf16')
        self.wire = torch.empty(grads.numel(), dtype=torch.bfloat16, device=grads.device) if grad_dtype == 'bf16' else None
        self.timing = bool(timing) and grads.is_cuda
        self.timed = []              # per bucket of the last step: [start, done, consumer-wait] events
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if always_reduce and dist.is_initialized():
            self.world = max(self.world, 2)          # only ever compared with 1: "there is a collective to run"
        self.buckets = make_buckets(offsets_sizes, grads.numel(), int(bucket_mb * 1024 * 1024 / 4))
        self.use_stream = grads.is_cuda
        # high priority: a bucket's collective should start as soon as its gradients are final, not queue behind backward kernels
        self.comm_stream = torch.cuda.Stream(priority=-1) if self.use_stream else None
        self.next_bucket = 0
        self.handles = []
        self.pending_events = []
        self.done_events = []
        self.params = shard_params
        self.shard = shard_params is not None
        self.rank = (dist.get_rank(group) if dist.is_initialized() else 0) if rank is None else rank
        self.update_fn = None
        # reduce-scatter / all-gather into one tensor exist on RCCL ("nccl"); every other backend takes the all-reduce emulation
        self.native_shard_collectives = bool(dist.is_initialized() and dist.get_backend(group) == 'nccl')

    def begin(self):
        self.next_bucket = 0
        self.handles = []
        self.pending_events = []
        self.done_events = []        # per launched bucket: completes when its all-reduce has (CUDA path)
        self.timed = []

    def ready_upto(self, end_offset, event=None):
        """Backward has enqueued every gradient in [0, end_offset): launch the complete buckets.  ``event`` (optional)
        completes when gradients enqueued on another stream (the engine's filter-gradient stream) are final; the
        communication stream waits for every such event seen so far plus the caller's current stream."""
        if event is not None:
            self.pending_events.append(event)
        if self.world == 1:
            return
        while self.next_bucket < len(self.buckets) and self.buckets[self.next_bucket][1] <= end_offset:
            s, e = self.buckets[self.next_bucket]
            self._launch(s, e)
            self.next_bucket += 1

    # ---- optimizer sharding: reduce-scatter -> update own shard -> all-gather, per bucket
    def _reduce_scatter(self, buf, s, own_lo, own_hi, send):
        """buf[own_lo:own_hi] = sum over ranks of their buf[own_lo:own_hi]; [s, send) = world equal shards.  In place (the output is this
        rank's slice of the input).  The path is chosen ONCE from the group's backend (``self.native_shard_collectives``), never from a caught
        exception: a rank-local RCCL error must propagate (and be agreed across ranks by the caller, like every other rank-local
        error), not make that one rank issue a different collective than its peers.  Backends without reduce-scatter into a tensor
        (gloo: the CPU tests) all-reduce the range -- same values."""
        if send == s:
            return
        if self.native_shard_collectives:
            dist.reduce_scatter_tensor(buf[own_lo:own_hi], buf[s:send], op=dist.ReduceOp.SUM, group=self.group)
        else:
            dist.all_reduce(buf[s:send], op=dist.ReduceOp.SUM, group=self.group)

    def _all_gather(self, buf, s, own_lo, own_hi, send):
        if send == s:
            return
        if self.native_shard_collectives:
            dist.all_gather_into_tensor(buf[s:send], buf[own_lo:own_hi], group=self.group)
        else:
            # gloo (tests): no all-gather into one tensor, none at all for device tensors -- sum of zero-padded shards instead
            tmp = torch.zeros_like(buf[s:send])
            tmp[own_lo - s:own_hi - s].copy_(buf[own_lo:own_hi])
            dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group)
            buf[s:send].copy_(tmp)

    def _sharded_chain(self, s, e):
        """On the current stream (the communication stream on a GPU): exchange, update, gather of one bucket."""
        from . import ops
        world = dist.get_world_size(self.group)
        own_lo, own_hi, send = shard_of(s, e, world, self.rank)
        g = self.grads
        if self.wire is not None and g.is_cuda:
            ops.cast_f32_bf16(g[s:e], self.wire[s:e], e - s)
            self._reduce_scatter(self.wire, s, own_lo, own_hi, send)
            if send < e:
                dist.all_reduce(self.wire[send:e], op=dist.ReduceOp.SUM, group=self.group)
            if own_hi > own_lo:
                ops.cast_bf16_f32(self.wire[own_lo:own_hi], g[own_lo:own_hi], own_hi - own_lo)
            if send < e:
                ops.cast_bf16_f32(self.wire[send:e], g[send:e], e - send)
        else:
            if self.wire is not None:          # host tensors: torch's converting copies stand in for the cast kernels
                self.wire[s:e].copy_(g[s:e])
                self._reduce_scatter(self.wire, s, own_lo, own_hi, send)
                if send < e:
                    dist.all_reduce(self.wire[send:e], op=dist.ReduceOp.SUM, group=self.group)
                g[own_lo:own_hi].copy_(self.wire[own_lo:own_hi])
                g[send:e].copy_(self.wire[send:e])
            else:
                self._reduce_scatter(g, s, own_lo, own_hi, send)
                if send < e:
                    dist.all_reduce(g[send:e], op=dist.ReduceOp.SUM, group=self.group)
        assert self.update_fn is not None, 'optimizer sharding: the session sets update_fn before backward'
        if own_hi > own_lo:
            self.update_fn(own_lo, own_hi)
        if send < e:
            self.update_fn(send, e)             # (the replicated remainder: < world * 64 elements)
        self._all_gather(self.params, s, own_lo, own_hi, send)

    def gather_slots(self, slots):
        """Optimizer sharding: every rank holds valid optimizer state for its own shards only; before a checkpoint (or a switch back to
        the replicated update) the shards of every slot arena are all-gathered so that each rank holds the complete state."""
        if not self.shard or self.world == 1 or not dist.is_initialized():
            return
        world = dist.get_world_size(self.group)
        for s, e in self.buckets:
            own_lo, own_hi, send = shard_of(s, e, world, self.rank)
            for slot in slots:
                self._all_gather(slot, s, own_lo, own_hi, send)

    def _launch(self, s, e):
        view = self.grads[s:e]
        if self.shard:
            if self.use_stream:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                self.comm_stream.wait_event(ev)
                for pe in self.pending_events:
                    self.comm_stream.wait_event(pe)
                self.pending_events = []
                with torch.cuda.stream(self.comm_stream):
                    if self.timing:
                        start = torch.cuda.Event(enable_timing=True)
                        start.record(self.comm_stream)
                    self._sharded_chain(s, e)
                    done = torch.cuda.Event(enable_timing=self.timing)
                    done.record(self.comm_stream)
                self.done_events.append(done)
                if self.timing:
                    self.timed.append([start, done, None, (e - s) * (2 if self.wire is not None else 4) + (e - s) * 4])
            else:
                self._sharded_chain(s, e)
                self.handles.append((None, s, e))
            return
        if self.use_stream:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            for pe in self.pending_events:
                self.comm_stream.wait_event(pe)
            self.pending_events = []
            with torch.cuda.stream(self.comm_stream):
                if self.timing:
                    start = torch.cuda.Event(enable_timing=True)
                    start.record(self.comm_stream)
                if self.wire is not None:
                    from . import ops
                    ops.cast_f32_bf16(view, self.wire[s:e], e - s)          # (on the communication stream: ops launch on the current stream)
                    dist.all_reduce(self.wire[s:e], op=dist.ReduceOp.SUM, group=self.group)
                    ops.cast_bf16_f32(self.wire[s:e], view, e - s)
                else:
                    dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
                done = torch.cuda.Event(enable_timing=self.timing)
                done.record(self.comm_stream)
            self.done_events.append(done)
            if self.timing:
                self.timed.append([start, done, None, (e - s) * (2 if self.wire is not None else 4)])
        elif self.wire is not None:
            # host tensors (gloo tests): the same protocol, with torch's dtype-converting copies standing in for the two cast kernels
            self.wire[s:e].copy_(view)
            self.handles.append((dist.all_reduce(self.wire[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True), s, e))
        else:
            self.handles.append((dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True), s, e))

    def _wait_host(self, i):
        h, s, e = self.handles[i]
        if h is not None:
            h.wait()
            if self.wire is not None:
                self.grads[s:e].copy_(self.wire[s:e])
            self.handles[i] = (None, s, e)

    def finish(self, wait=True):
        """Flushes the remaining buckets; with ``wait`` the compute stream then waits for every collective.  With
        wait=False the caller consumes the buckets one by one through ``completed_buckets()`` (the optimizer updates bucket k
        while buckets k+1.. are still on the wire)."""
        if self.world == 1:
            return
        self.ready_upto(self.grads.numel())
        if not wait:
            return
        if self.use_stream:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            for i in range(len(self.handles)):
                self._wait_host(i)

    def completed_buckets(self):
        """Yields (start, end) of every bucket in launch order, each after making the current stream (or the host, on the CPU
        path) wait for that bucket's all-reduce only."""
        for i, (s, e) in enumerate(self.buckets):
            if self.world > 1:
                if self.use_stream:
                    if self.timing:
                        w = torch.cuda.Event(enable_timing=True)
                        w.record(torch.cuda.current_stream())
                        self.timed[i][2] = w
                    torch.cuda.current_stream().wait_event(self.done_events[i])
                else:
                    self._wait_host(i)
            yield s, e

    def exposed_times(self):
        """After a synchronised step run with ``timing``: per bucket {'bytes' on the wire, 'collective_ms' (cast + all-reduce + cast on the
        communication stream), 'exposed_ms' = how long the consumer's stream had to wait for it (0 when the bucket had already landed)}."""
        out = []
        for start, done, wait, nbytes in self.timed:
            exposed = max(0.0, wait.elapsed_time(done)) if wait is not None else None
            out.append({'bytes': int(nbytes), 'collective_ms': start.elapsed_time(done), 'exposed_ms': exposed})
        return out

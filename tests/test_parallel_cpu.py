"""Data-parallel path on CPU: bucket construction, the reverse-order arena layout that makes buckets
complete front-to-back during backward, and a world_size-2 gloo run of GradReducer (the same code
path that drives RCCL on the GPUs, minus the side stream)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _graph(inference='darknet'):
    from yolo_tf_amd import graph as G
    from yolo_tf_amd.model.yolo2 import inference as inf
    g = G.Graph()
    x = G.placeholder(g, 'image', 416, 416)
    getattr(inf, inference)(x, 20, 5, training=True)
    return g


def test_arena_layout_is_reverse_creation_order_and_backward_monotone():
    from yolo_tf_amd.engine import layout_params, layer_end_offsets
    g = _graph()
    offsets, total = layout_params(g)
    from yolo_tf_amd.engine import ARENA_ALIGN
    assert total >= 67_100_000 and total % ARENA_ALIGN == 0
    spans = sorted(offsets.values())
    assert spans[0][0] == 0
    for (o1, n1), (o2, _) in zip(spans, spans[1:]):
        assert o1 + (n1 + ARENA_ALIGN - 1) // ARENA_ALIGN * ARENA_ALIGN == o2          # contiguous up to the 256-byte alignment of every variable
    # the last layer's variables come first: their gradients are the first to be final
    assert offsets['yolo2_darknet/conv/biases'][0] < offsets['yolo2_darknet/conv20/weights'][0] < offsets['yolo2_darknet/conv0/weights'][0]
    ends = layer_end_offsets(g, offsets)
    seq = [ends[op['name']] for op in reversed(g.ops) if op['kind'] == 'conv']
    assert seq == sorted(seq) and seq[-1] == total


def test_buckets_partition_the_arena_at_variable_boundaries():
    from yolo_tf_amd.engine import layout_params
    from yolo_tf_amd.parallel import make_buckets
    offsets, total = layout_params(_graph())
    for mb in (1, 16, 64, 1024):
        buckets = make_buckets(offsets.values(), total, int(mb * 1024 * 1024 / 4))
        assert buckets[0][0] == 0 and buckets[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(buckets, buckets[1:]))
        bounds = {o for o, n in offsets.values()} | {total}
        assert all(e in bounds for _, e in buckets)
        assert all(e - s >= mb * 1024 * 1024 / 4 for s, e in buckets[:-1])
    assert len(make_buckets(offsets.values(), total, 16 * 1024 * 1024)) >= 3   # 64 MiB buckets: several all-reduces in flight


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from yolo_tf_amd.parallel import GradReducer, init_distributed
    r, _, w = init_distributed(backend='gloo')
    assert (r, w) == (rank, world)
    sizes = [400, 250, 7, 1000, 364, 3]                     # variables of a toy arena, every one 4-aligned in the arena
    offs, off = [], 0
    for n in sizes:
        offs.append((off, n))
        off += (n + 3) // 4 * 4
    grads = torch.arange(off, dtype=torch.float32) * (rank + 1)
    red = GradReducer(grads, offs, bucket_mb=300 * 4 / (1024 * 1024))          # ~300-element buckets -> several buckets
    assert len(red.buckets) >= 3
    for step in range(2):                                   # reusable across steps
        if step:
            grads.copy_(torch.arange(off, dtype=torch.float32) * (rank + 1))
        red.begin()
        launched = []
        for end in sorted(o + (n + 3) // 4 * 4 for o, n in offs):      # backward finishing layer after layer
            red.ready_upto(end)
            launched.append(red.next_bucket)
        assert launched == sorted(launched)
        red.finish()
        expect = torch.arange(off, dtype=torch.float32) * sum(range(1, world + 1))
        assert torch.equal(grads, expect), (rank, step)
    # deferred consumption: finish(wait=False) + completed_buckets() hands the buckets over one by one, each already summed
    grads.copy_(torch.arange(off, dtype=torch.float32) * (rank + 1))
    red.begin()
    red.finish(wait=False)
    seen = []
    for lo, hi in red.completed_buckets():
        assert torch.equal(grads[lo:hi], torch.arange(lo, hi, dtype=torch.float32) * sum(range(1, world + 1))), (rank, lo, hi)
        seen.append((lo, hi))
    assert seen == red.buckets
    # folded averaging: the optimizer's gscale = 1/world turns the sum into the mean
    assert torch.allclose(grads / world, torch.arange(off, dtype=torch.float32) * (world + 1) / 2)
    # bf16 wire format (half the bytes per link): the same bucket protocol; each rank's contribution is rounded to bf16, the sum is
    # formed in bf16 and widened back into the f32 arena -- equal on every rank, within bf16 rounding of the exact sum
    g2 = (torch.arange(off, dtype=torch.float32) * 0.37 + 1.0) * (rank + 1)
    red16 = GradReducer(g2, offs, bucket_mb=300 * 4 / (1024 * 1024), grad_dtype='bf16')
    assert red16.wire.dtype == torch.bfloat16 and red16.buckets == red.buckets
    red16.begin()
    red16.finish(wait=False)
    for lo, hi in red16.completed_buckets():
        exact = (torch.arange(lo, hi, dtype=torch.float32) * 0.37 + 1.0) * sum(range(1, world + 1))
        assert torch.all((g2[lo:hi] - exact).abs() <= 2 ** -7 * exact.abs()), (rank, lo, hi)
        assert torch.equal(g2[lo:hi], g2[lo:hi].to(torch.bfloat16).float())          # what arrived is a bf16 value
    gathered = [torch.empty_like(g2) for _ in range(world)]
    dist.all_gather(gathered, g2)
    assert all(torch.equal(gathered[0], t) for t in gathered)                        # replicas see identical gradients
    if rank == 0:
        open(out, 'w').write('ok')
    dist.barrier()
    dist.destroy_process_group()


def test_grad_reducer_gloo_world2(tmp_path):
    out = str(tmp_path / 'ok')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == 'ok'


def _shard_worker(rank, world, port, out, grad_dtype):
    """Optimizer sharding on host tensors: reduce-scatter / update 1/world / all-gather per bucket == all-reduce + replicated update."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from yolo_tf_amd.parallel import GradReducer, init_distributed, shard_of
    init_distributed(backend='gloo')
    sizes = [4000, 2500, 70, 10000, 3640, 30, 129]          # a toy arena: variables on 64-element boundaries like engine.layout_params
    offs, off = [], 0
    for n in sizes:
        offs.append((off, n))
        off += (n + 63) // 64 * 64
    bucket_mb = 3000 * 4 / (1024 * 1024)

    def run(shard):
        torch.manual_seed(5)
        params = torch.randn(off)
        m = torch.zeros(off)
        grads = torch.empty(off)
        red = GradReducer(grads, offs, bucket_mb=bucket_mb, grad_dtype=grad_dtype, shard_params=params if shard else None)
        assert len(red.buckets) >= 3

        def update(lo, hi):            # a stateful elementwise optimizer (momentum), like the kernels: touches [lo, hi) only
            m[lo:hi].mul_(0.9).add_(grads[lo:hi] / world)
            params[lo:hi].sub_(0.1 * m[lo:hi])
        for step in range(3):
            g = torch.Generator().manual_seed(100 * step + rank)
            grads.copy_(torch.randn(off, generator=g))
            red.begin()
            red.update_fn = update
            for end in sorted(o + (n + 63) // 64 * 64 for o, n in offs):
                red.ready_upto(end)
            if shard:
                red.finish(wait=True)                       # chains ran inside ready_upto: exchange, shard update, gather
            else:
                red.finish(wait=False)
                for lo, hi in red.completed_buckets():
                    update(lo, hi)
        if shard:
            # slots are valid for the rank's own shards (+ the replicated remainders) until gathered
            own = torch.zeros(off, dtype=torch.bool)
            for s_, e_ in red.buckets:
                lo, hi, send = shard_of(s_, e_, world, rank)
                own[lo:hi] = True
                own[send:e_] = True
            red.gather_slots([m])
        return params, m

    p_rep, m_rep = run(False)
    p_sh, m_sh = run(True)
    assert torch.equal(p_sh, p_rep), (rank, float((p_sh - p_rep).abs().max()))        # bit-identical parameters (two ranks: a+b == b+a)
    assert torch.equal(m_sh, m_rep)
    gathered = [torch.empty_like(p_sh) for _ in range(world)]
    dist.all_gather(gathered, p_sh)
    assert all(torch.equal(gathered[0], t) for t in gathered)                         # replicas stay identical
    lo, hi, send = shard_of(0, 1000, 2, rank)
    assert (hi - lo) == 448 and send == 896 and lo == rank * 448                      # 1000 = 2 x 448 + a replicated remainder of 104
    if rank == 0:
        open(out, 'w').write('ok')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('grad_dtype', ['f32', 'bf16'])
def test_optimizer_sharding_equals_replicated_update_gloo_world2(tmp_path, grad_dtype):
    """[mi355x] shard_optimizer (parallel.GradReducer shard_params): after three steps the parameters and the (gathered) optimizer state are
    bit-identical to the all-reduce + replicated-update path, on both ranks, with f32 and bf16 wire formats."""
    out = str(tmp_path / 'ok')
    mp.spawn(_shard_worker, args=(2, _free_port(), out, grad_dtype), nprocs=2, join=True)
    assert open(out).read() == 'ok'


def test_shard_collective_errors_propagate_instead_of_switching_collectives(monkeypatch):
    """The reduce-scatter / all-gather path is chosen from the backend, once; an error raised by the collective itself (a rank-local RCCL
    failure) must reach the caller -- it used to be caught and answered with an all-reduce on that rank only, while its peers sat in
    reduce-scatter: mismatched collectives, a hang instead of a message."""
    from yolo_tf_amd import parallel
    g = torch.ones(256)
    red = parallel.GradReducer(g, [(0, 256)], bucket_mb=1, shard_params=torch.zeros(256), rank=0)
    assert red.native_shard_collectives is False                 # no process group / gloo: the all-reduce emulation
    calls = []

    def boom(*a, **k):
        calls.append('native')
        raise RuntimeError('NCCL error: unhandled system error (injected)')
    monkeypatch.setattr(parallel.dist, 'reduce_scatter_tensor', boom)
    monkeypatch.setattr(parallel.dist, 'all_gather_into_tensor', boom)
    monkeypatch.setattr(parallel.dist, 'all_reduce', lambda *a, **k: calls.append('all_reduce'))
    red.native_shard_collectives = True                          # what dist.get_backend(group) == 'nccl' selects
    with pytest.raises(RuntimeError, match='injected'):
        red._reduce_scatter(g, 0, 0, 128, 256)
    with pytest.raises(RuntimeError, match='injected'):
        red._all_gather(g, 0, 0, 128, 256)
    assert calls == ['native', 'native']                         # no silent second collective


def test_single_process_reducer_is_a_noop():
    from yolo_tf_amd.parallel import GradReducer
    g = torch.ones(100)
    red = GradReducer(g, [(0, 100)], bucket_mb=1)
    red.begin()
    red.ready_upto(100)
    red.finish()
    assert torch.equal(g, torch.ones(100))


def test_host_side_helpers():
    """config overlay order, cell-grid asserts, lr schedule -- host logic shared by train.py / detect.py."""
    import configparser
    from yolo_tf_amd import utils
    from yolo_tf_amd.optim import learning_rate_fn
    cfg = utils.make_config([os.path.join(ROOT, 'config.ini'), os.path.join(ROOT, 'config', 'yolo2', 'darknet-20.ini')], '/tmp/base')
    assert cfg.get('cache', 'names') == 'config/names/20' and cfg.get('yolo2', 'anchors').endswith('voc.tsv')   # later file wins
    assert utils.get_logdir(cfg) == '/tmp/base/yolo2/darknet/20' and utils.get_cachedir(cfg) == '/tmp/base/cache/20'
    assert utils.calc_cell_width_height(cfg, 416, 416) == (13, 13) and utils.get_downsampling(cfg) == (32, 32)
    with pytest.raises(AssertionError):
        utils.calc_cell_width_height(cfg, 400, 416)
    lr = learning_rate_fn(cfg, 1.0)
    assert lr(0) == 1.0 and lr(99999) == 1.0 and abs(lr(100000) - 0.96) < 1e-12 and abs(lr(250000) - 0.96 ** 2) < 1e-12
    cfg.remove_section('exponential_decay')
    assert learning_rate_fn(cfg, 0.5)(10 ** 6) == 0.5
    anchors = utils.read_anchors(os.path.join(ROOT, 'config', 'yolo2', 'anchors', 'voc.tsv'))
    assert anchors.shape == (5, 2) and np.allclose(anchors[0], [1.08, 1.19])

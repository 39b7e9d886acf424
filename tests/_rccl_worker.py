"""Worker of tests/test_rccl_gpu.py: ONE rank, backend "nccl" (= RCCL on ROCm), the production GradReducer path."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
# two forward passes are compared below: the standalone statistics pass is order-independent (see test_network_gpu._dp_worker)
os.environ['YOLO2_FUSE_BN_STATS'] = '0'

import numpy as np
import torch
import torch.distributed as dist


def main():
    torch.cuda.set_device(0)
    dist.init_process_group(backend='nccl', rank=0, world_size=1)
    assert dist.get_backend() == 'nccl'
    from test_network_gpu import make_builder
    from yolo_tf_amd.parallel import GradReducer, sync_replicas
    from yolo_tf_amd.session import TrainSession
    from yolo_tf_amd.utils import data
    from yolo_tf_amd import ops
    # 1. a bucketed all-reduce of a flat arena on the communication stream
    g = torch.arange(3_000_000, dtype=torch.float32, device='cuda')
    red = GradReducer(g, [(0, 1_000_000), (1_000_000, 1_000_000), (2_000_000, 1_000_000)], bucket_mb=2.0, always_reduce=True)
    assert len(red.buckets) >= 2
    red.begin()
    red.ready_upto(1_000_000)
    red.finish(wait=True)
    torch.cuda.synchronize()
    assert torch.equal(g, torch.arange(3_000_000, dtype=torch.float32, device='cuda'))     # SUM over one rank
    # 2. the whole data-parallel training step with its collectives routed through RCCL
    b, _ = make_builder('tiny', 20, 96, True, tempfile.mkdtemp(prefix='yolo_rccl_'))
    sess = TrainSession(b, 2, dtype='f32', optimizer='adam', learning_rate=1e-3, seed=3)
    e = sess.engine
    rng = np.random.RandomState(0)
    images = torch.from_numpy(rng.uniform(0, 255, (2, 96, 96, 3)).astype(np.float32)).cuda()
    sess.upload_labels(data.synthetic_batch(2, 20, 3, 3, seed=1))
    sess.forward_backward(images)
    local = e.grads.clone()
    sess.reducer = GradReducer(e.grads, list(e.param_offsets.values()), 8.0, always_reduce=True)
    assert len(sess.reducer.buckets) >= 3
    sess.forward_backward(images, defer_collectives=True)
    sess.apply_gradients()                      # consumes the buckets one by one as their all-reduces complete
    torch.cuda.synchronize()
    assert len(sess.reducer.done_events) == len(sess.reducer.buckets)
    # f32 atomics: the two backward passes agree to rounding, and the RCCL sum over one rank changes nothing
    assert float((e.grads - local).abs().max()) <= 1e-5 * float(local.abs().max())
    # 3. the same step with the bf16 wire format and per-bucket timing: gradients equal the local ones up to bf16 rounding (same weights:
    #    no update in between), and the exposed-time report (what bench.py prints for N > 1) has one entry per bucket
    sess.reducer = None
    sess.forward_backward(images)
    local = e.grads.clone()
    sess.reducer = GradReducer(e.grads, list(e.param_offsets.values()), 8.0, always_reduce=True, grad_dtype='bf16', timing=True)
    sess.forward_backward(images)               # collectives waited for on return
    torch.cuda.synchronize()
    assert float((e.grads - local).abs().max()) <= 2 ** -7 * float(local.abs().max())
    assert torch.equal(e.grads, e.grads.to(torch.bfloat16).float())
    sess.forward_backward(images, defer_collectives=True)
    sess.apply_gradients()
    torch.cuda.synchronize()
    rep = sess.reducer.exposed_times()
    assert len(rep) == len(sess.reducer.buckets) and all(r['collective_ms'] > 0 and r['exposed_ms'] is not None and r['exposed_ms'] >= 0 for r in rep)
    assert sum(r['bytes'] for r in rep) == 2 * e.grads.numel()
    sess.global_step = 1
    sync_replicas(sess, always=True)            # broadcast path (parameters, statistics, slots, global_step)
    assert sess.global_step == 1
    assert ops.get_stream_workgroups() == torch.cuda.get_device_properties(0).multi_processor_count      # (a one-rank session does not reserve CUs)
    dist.barrier()
    dist.destroy_process_group()
    print('RCCL_OK ranks=1 buckets=%d' % len(sess.reducer.buckets))


if __name__ == '__main__':
    main()

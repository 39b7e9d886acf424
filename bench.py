#!/usr/bin/env python
"""bench.py -- training throughput of the MI355X YOLOv2 hot path on BASELINE.json's metric:
img/s training Darknet-19 YOLOv2 VOC-20 416x416 bf16, batch 16 per GPU (configs[1]); weak scaling
over N GPUs of one node (one process per GPU, RCCL gradient all-reduce overlapped with backward).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8                       (re-executes itself under torch.distributed.run: one rank per GPU, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus 8 --global-batch 64 --names 80     (BASELINE configs[2]: strong scaling, 8 images per GPU)
    python bench.py --gpus 8 --shard-optimizer                 (reduce-scatter / 1/N update / all-gather instead of all-reduce + replicated update)
    python bench.py --multiscale [--gpus 4]                    (BASELINE configs[3]: a different input size every step)

A "step" = per-image standardisation -> forward (batch-stat BN) -> YOLOv2 loss fwd+bwd -> backward
-> gradient all-reduce (N>1) -> Adam, on a synthetic batch that is already resident in HBM.
Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      the dominant kernel: the 24 3x3 forward + data-gradient launches with > 64 filters (conv3x3_pp_kernel on images up to
                55 wide, conv_igemm_kernel<bf16,128,...,KS=3> on the 104x104 stage) -- algorithmic FLOPs / RAW HIP-event time of
                those launches in instrumented steps right after the timed region, against the 2.5 PFLOP/s dense bf16 MFMA
                peak; the bracket-calibrated figures and the PMC traffic constant (profiles/dominant_kernel_pmc.json) ride along
  cpu_baseline  value = the torch-CPU (oneDNN) forward+backward of the same conv stack at batch 8 on the host cores (rank 0, N=1
                only; oracle/torch_cpu_ref.py, a port: TF-1.0 cannot run here); sub-fields: the NumPy oracle's full training
                step on one image (oracle/yolo2_ref.py) and the single-thread C restatement of the reference NMS
  f32_parity_mode  the same step in the reference's own precision (exact-f32 MFMA), 10 steps outside the timed region: img/s and the
                fraction of the 157.3 TFLOP/s f32 matrix peak
  detect        BASELINE configs[4]: batch-256 detect p50/p99 with (a) the network's own scores, (b) the sparse and
                (c) the dense NMS stress inputs of BASELINE.md section 2 written over the decoded boxes
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np   # noqa: E402
import torch         # noqa: E402

BF16_DENSE_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
F32_MATRIX_PEAK_TFLOPS = 157.3
# Fabric bytes per launch of the dominant kernel (the 24 3x3 launches of one step): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE
# in separate passes over this same command (scripts/gpu_traffic.sh), FETCH doubled per the guide's gfx950 note, WRITE calibrated
# 1.00 on bn_leaky_kernel (profiles/r01_hbm_traffic_pmc_final.md: 99.1 MB fetched + 27.2 MB written per launch).  The counters
# sit between L2 and the fabric: Infinity-Cache hits are included (the 3072-channel layer alone re-reads its filter slab from
# the MALL 11 times: 0.8 GB).  Algorithmic bytes (every operand once): 31.9 MB, 40.4 GFLOP per launch.
# The two figures live in a tracked profile file together with the commit they were measured at (they go stale with every kernel change;
# a constant pasted here would hide that): profiles/dominant_kernel_pmc.json, written from scripts/gpu_traffic.sh runs.
PMC_PROFILE = os.path.join(ROOT, 'profiles', 'dominant_kernel_pmc.json')
with open(PMC_PROFILE) as _f:
    PMC = json.load(_f)
IGEMM_HBM_BYTES_PER_LAUNCH = float(PMC['traffic_bytes_per_launch'])
IGEMM_ALGORITHMIC_BYTES_PER_LAUNCH = float(PMC['algorithmic_bytes_per_launch'])
TRAIN_GFLOP_PER_IMG = {20: 104.396, 80: 104.707}    # 2*(3*sum(MACs) - MACs(conv0))


class KernelTimer(object):
    """HIP events around every launch of the dominant kernel, on the stream it is launched on."""

    def __init__(self):
        self.pairs = []
        self.flops = []
        self.tags = []
        self.enabled = False
        self._cur = None

    def start(self, flops, tag='fwd'):
        if not self.enabled:
            return
        a = torch.cuda.Event(enable_timing=True)
        a.record(torch.cuda.current_stream())
        self._cur = (a, flops, tag)

    def stop(self):
        if self._cur is None:
            return
        b = torch.cuda.Event(enable_timing=True)
        b.record(torch.cuda.current_stream())
        self.pairs.append((self._cur[0], b))
        self.flops.append(self._cur[1])
        self.tags.append(self._cur[2])
        self._cur = None

    NOOP_KERNEL_MS = float(PMC['empty_kernel_duration_ms'])      # median duration rocprofv3 reports for the empty kernel itself (64 launches: 0.6 .. 6.9 us)

    def calibrate(self, n=64):
        """What a bracket adds to the kernel inside it.  A HIP-event pair measures from the completion of the start event to the
        completion of the stop event: the bracketed kernel's own duration (what rocprofv3's kernel trace reports) PLUS the dispatch
        latency between the two (~3 us; more under a profiler's dispatch interception).  Measured on brackets around an empty kernel,
        each issued behind a running kernel like the real ones: overhead = median bracket - the empty kernel's own 3.4 us."""
        from yolo_tf_amd import ops
        st = torch.cuda.current_stream()
        filler = torch.zeros(1 << 22, device='cuda')
        pairs = []
        for _ in range(n):
            filler.add_(1.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            ops.noop()
            b.record(st)
            pairs.append((a, b))
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in pairs)
        self.bracket_noop_ms = ms[len(ms) // 2]
        self.bracket_overhead_ms = max(0.0, self.bracket_noop_ms - self.NOOP_KERNEL_MS)

    def summary(self, tag=None):
        tags = None if tag is None else (tag if isinstance(tag, tuple) else (tag,))
        sel = [i for i in range(len(self.pairs)) if tags is None or self.tags[i] in tags]
        if not sel:
            return None
        raw = [self.pairs[i][0].elapsed_time(self.pairs[i][1]) for i in sel]
        ovh = getattr(self, 'bracket_overhead_ms', 0.0)
        ms = [max(m - ovh, 0.0) for m in raw]
        fl = [self.flops[i] for i in sel]
        total_ms = float(sum(ms))
        return {'launches': len(ms), 'avg_ms': total_ms / len(ms), 'total_ms': total_ms, 'raw_bracket_avg_ms': float(sum(raw)) / len(raw),
                'tflops': float(sum(fl)) / (total_ms * 1e-3) / 1e12, 'flop_per_launch': float(sum(fl)) / len(ms)}


def make_builder(inference, names, size, training, basedir):
    from yolo_tf_amd import utils
    from yolo_tf_amd.model import yolo2
    cfg = utils.make_config([os.path.join(ROOT, 'config.ini'), os.path.join(ROOT, 'config', 'yolo2', '%s-%d.ini' % (inference, names))], basedir)
    cfg.set('cache', 'names', os.path.join(ROOT, cfg.get('cache', 'names')))
    cfg.set('yolo2', 'anchors', os.path.join(ROOT, cfg.get('yolo2', 'anchors')))
    cfg.set('yolo2', 'width', str(size))
    cfg.set('yolo2', 'height', str(size))
    utils.ensure_names(cfg)
    b = yolo2.Builder(None, cfg)
    b(None, training=training)
    if training:
        b.create_objectives()
    return b, cfg


def cpu_baseline(names, size, budget_s=30.0):
    """Times the NumPy oracle's train step (forward + loss + backward + Adam) on ONE image of the same
    workload shape.  Test infrastructure used as the reported CPU baseline; never on the product path."""
    from oracle import yolo2_ref as R
    from yolo_tf_amd.utils import data
    anchors = np.loadtxt(os.path.join(ROOT, 'config', 'yolo2', 'anchors', 'voc.tsv' if names == 20 else 'coco.tsv'), delimiter='\t', skiprows=1)
    spec = R.darknet_spec(names, len(anchors))
    params = R.init_params(spec, seed=0)
    rng = np.random.RandomState(0)
    cells = size // 32
    hp = {'prob': 1., 'iou_best': 5., 'iou_normal': 1., 'coords': 1.}
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count()
    n, t_total = 0, 0.0
    while n < 1 or (t_total < 10.0 and t_total / n * (n + 1) < budget_s):
        x = rng.randn(1, size, size, 3).astype(np.float32)
        labels = data.synthetic_batch(1, names, cells, cells, seed=n)
        t0 = time.time()
        R.train_step(spec, params, {}, x, labels, names, anchors, hp, 1e-6, 0)
        t_total += time.time() - t0
        n += 1
    port = {'value': n / t_total, 'unit': 'img/s', 'cores': int(threads),
            'sample': '%d single-image 416x416 Darknet-19 training steps (fwd+loss+bwd+Adam) of oracle/yolo2_ref.py, NumPy/BLAS f32, %.1f s; '
                      'CPU restatement of reference semantics (TensorFlow 1.0 unavailable)' % (n, t_total)}
    out = dict(port, kind='port', host_cores=os.cpu_count())
    # conv stack on torch-CPU (oneDNN, NHWC f32) at the reference's default batch 8 (train.py:156) -- the closest stand-in for
    # TF-1.0's CPU kernels this image can run, and the faster of the two CPU legs: it is the headline `value`; the NumPy port stays beside it
    try:
        from oracle import torch_cpu_ref as T
        r = T.time_conv_stack(classes=names, size=size, batch=8, budget_s=12.0)
        out['torch_cpu_conv_stack'] = {'fwd_img_s': r['fwd_img_s'], 'train_img_s': r['train_img_s'], 'batch': 8, 'threads': r['threads'],
                                       'sample': '%d forward / %d forward+backward passes, Darknet-19 %dx%d f32 channels_last' % (r['fwd_iters'], r['train_iters'], size, size)}
        out.update({'value': r['train_img_s'], 'cores': int(r['threads']), 'numpy_port': port,
                    'sample': '%d forward+backward passes of the Darknet-19 conv stack at batch 8, %dx%d f32 channels_last, torch-CPU / oneDNN '
                              '(oracle/torch_cpu_ref.py: a CPU port of the same layer stack; no loss / optimizer, which are < 1 %% of the step)'
                              % (r['train_iters'], size, size)})
    except Exception as exc:       # a baseline leg must never take the GPU numbers down with it
        out['torch_cpu_conv_stack'] = {'error': repr(exc)}
    # the reference's NMS algorithm (utils/postprocess.py:39-51), single thread like the reference, C restatement
    try:
        out['nms_ms_per_image'] = cpu_nms_baseline(names)
    except Exception as exc:
        out['nms_ms_per_image'] = {'error': repr(exc)}
    return out


def nms_stress_inputs(kind, batch, classes, seed0=0):
    """The NMS stress inputs of BASELINE.md section 2 (tests/golden/make_golden.py recipes), one seed per image:
    'sparse': background conf U(0, 0.05), 12 boxes at 0.5-0.9, wh U(0.5, 5.5); 'dense': conf U(0, 0.5), wh U(0, 4) -- the
    reference's worst case (69.5 s per image in its Python loop).  Returns conf [B,845,C], xy_min, xy_max [B,845,2] (f32)."""
    conf = np.zeros((batch, 845, classes), np.float32)
    mn = np.zeros((batch, 845, 2), np.float32)
    mx = np.zeros((batch, 845, 2), np.float32)
    for i in range(batch):
        rng = np.random.RandomState(seed0 + i)
        if kind == 'sparse':
            c = rng.uniform(0, 0.05, (845, classes)).astype(np.float32)
            hot = rng.choice(845, 12, replace=False)
            c[hot, rng.randint(0, classes, 12)] = rng.uniform(0.5, 0.9, 12).astype(np.float32)
            cen, wh = rng.uniform(0, 13, (845, 2)), rng.uniform(0.5, 5.5, (845, 2))
        else:
            c = rng.uniform(0, 0.5, (845, classes)).astype(np.float32)
            cen, wh = rng.uniform(0, 13, (845, 2)), rng.uniform(0, 4, (845, 2))
        conf[i] = c
        mn[i] = (cen - wh / 2).astype(np.float32)
        mx[i] = (cen + wh / 2).astype(np.float32)
    return conf, mn, mx


def cpu_nms_baseline(classes):
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, 'oracle', 'libnms_ref.so'))
    P = ctypes.POINTER(ctypes.c_float)
    res = {'threads': 1, 'impl': 'oracle/nms_ref.c (C restatement of utils/postprocess.py:39-51; the reference itself is a Python loop: '
                                 '0.245 s sparse / 69.5 s dense per image, BASELINE.md)'}
    for kind, n_img in (('sparse', 8), ('dense', 4)):
        conf, mn, mx = nms_stress_inputs(kind, n_img, classes)
        order = np.zeros(845, np.int64)
        t0 = time.time()
        for i in range(n_img):
            lib.nms_ref(conf[i].ctypes.data_as(P), mn[i].ctypes.data_as(P), mx[i].ctypes.data_as(P), ctypes.c_long(845), ctypes.c_long(classes),
                        ctypes.c_float(0.3), ctypes.c_float(0.4), order.ctypes.data_as(ctypes.POINTER(ctypes.c_long)))
        res[kind] = (time.time() - t0) / n_img * 1e3
    return res


def f32_parity_mode(args, steps=10, warmup=3):
    """The reference's own arithmetic is fp32 (TF-1.0 float32 graph): the same training step in the f32 parity mode -- exact-f32 MFMA
    (v_mfma_f32_32x32x2_f32, 1/16 of the bf16 rate), the mode the 1e-4 oracle comparisons run in -- timed OUTSIDE the bench line's timed
    region, so that the parity mode has a driver-observed number beside the bf16 one."""
    from yolo_tf_amd.session import TrainSession
    from yolo_tf_amd.utils import data
    basedir = tempfile.mkdtemp(prefix='yolo_bench_f32_')
    builder, cfg = make_builder('darknet', args.names, args.size, True, basedir)
    sess = TrainSession(builder, args.batch, dtype='f32', optimizer='adam', learning_rate=1e-6, seed=0)
    cells = args.size // 32
    gen = torch.Generator(device='cuda').manual_seed(1234)
    images = torch.rand(args.batch, args.size, args.size, 3, device='cuda', generator=gen) * 255.0
    sess.upload_labels(data.synthetic_batch(args.batch, args.names, cells, cells, seed=4321))
    for _ in range(warmup):
        sess.step(images)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sess.step(images)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loss = sess.fetch()['total_loss']
    del sess
    torch.cuda.empty_cache()
    value = args.batch * steps / dt
    tflops = value * TRAIN_GFLOP_PER_IMG[args.names] * (args.size / 416.0) ** 2 / 1e3
    return {'value': value, 'unit': 'img/s', 'ms_per_step': dt / steps * 1e3, 'steps': steps, 'warmup': warmup, 'dtype': 'f32', 'batch': args.batch,
            'whole_step_tflops': tflops, 'whole_step_frac_of_f32_matrix_peak': tflops / F32_MATRIX_PEAK_TFLOPS, 'peak_tflops': F32_MATRIX_PEAK_TFLOPS,
            'total_loss': loss, 'note': 'same step and workload as the line itself in the f32 parity mode; outside the timed region'}


def _lib_env_overrides():
    from yolo_tf_amd import _lib
    return _lib.env_overrides()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100, help='timed steps (0.4 s of GPU time at the default: long enough for the clocks to settle; 20 / 100 / 300 steps measure the same rate within 1 %)')
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=16, help='images per GPU (weak scaling)')
    ap.add_argument('--global-batch', type=int, default=0, help='fixed total batch split over the GPUs (strong scaling; BASELINE configs[2] = 64 on 8)')
    ap.add_argument('--names', type=int, default=20, choices=[20, 80])
    ap.add_argument('--size', type=int, default=416)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--multiscale', action='store_true', help='BASELINE configs[3]: a different input size {320..608} every step (batch defaults to 8 per GPU)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timer', action='store_true')
    ap.add_argument('--grad-dtype', default=None, choices=['f32', 'bf16'], help='wire format of the gradient all-reduce (N > 1); default: [mi355x] grad_dtype')
    ap.add_argument('--shard-optimizer', action='store_true', help='N > 1: reduce-scatter + 1/N optimizer pass + all-gather instead of all-reduce + replicated update ([mi355x] shard_optimizer)')
    ap.add_argument('--backend', default=None, choices=['nccl', 'gloo'], help='process-group backend for N > 1 (default nccl = RCCL; gloo lets the tests run two ranks on one GPU)')
    ap.add_argument('--no-f32', action='store_true', help='skip the f32 parity-mode sub-object (10 steps of the same workload in exact-f32 arithmetic, rank 0, N = 1)')
    ap.add_argument('--no-detect', action='store_true', help='skip the batch-256 detect p50/p99 report (BASELINE configs[4]) on rank 0')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run (RCCL over xGMI)
        import socket
        import subprocess
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    from yolo_tf_amd.parallel import init_distributed
    import torch.distributed as dist
    if args.backend == 'gloo':       # (tests: N ranks sharing the visible GPUs)
        os.environ['LOCAL_RANK'] = str(int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
    rank, local_rank, world = init_distributed(args.backend)
    if world != args.gpus:
        raise SystemExit('bench.py --gpus %d was started with WORLD_SIZE=%d' % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    strong = args.global_batch > 0
    if strong:
        if args.global_batch % world:
            raise SystemExit('--global-batch %d is not divisible by %d GPUs' % (args.global_batch, world))
        args.batch = args.global_batch // world

    from yolo_tf_amd.session import TrainSession
    from yolo_tf_amd.utils import data
    basedir = tempfile.mkdtemp(prefix='yolo_bench_%d_' % rank)
    if args.multiscale:
        return multiscale(args, rank, world, basedir, dist)
    builder, cfg = make_builder('darknet', args.names, args.size, True, basedir)
    grad_dtype = args.grad_dtype or (cfg.get('mi355x', 'grad_dtype') if cfg.has_option('mi355x', 'grad_dtype') else 'f32')
    sess = TrainSession(builder, args.batch, dtype=args.dtype, optimizer='adam', learning_rate=1e-6, seed=0, world_size=world,
                        bucket_mb=cfg.getfloat('mi355x', 'bucket_mb'), grad_dtype=grad_dtype,
                        shard_optimizer=args.shard_optimizer or (cfg.has_option('mi355x', 'shard_optimizer') and cfg.getboolean('mi355x', 'shard_optimizer')))
    cells = args.size // 32
    gen = torch.Generator(device='cuda').manual_seed(1234 + rank)
    images = torch.rand(args.batch, args.size, args.size, 3, device='cuda', generator=gen) * 255.0
    sess.upload_labels(data.synthetic_batch(args.batch, args.names, cells, cells, seed=4321 + rank))
    timer = None if args.no_kernel_timer else KernelTimer()
    sess.engine.kernel_timer = timer

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        sess.step(images)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sess.step(images)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    # Roofline pass, right after the timed region on the same session: the same training steps with HIP events around
    # every launch of the dominant kernel.  Kept out of the timed region because a timed event is a barrier packet:
    # ~35 of them per step stop back-to-back dispatch and cost 11 % of the step (measured), which would understate
    # `value`.  The filter-gradient side stream is folded into the main stream for this pass so that each launch
    # runs alone and its event-to-event time is the kernel's own duration (as in the rocprofv3 single-stream summary).
    if timer:   # every rank steps (the optimizer all-reduces)
        overlap = sess.engine.overlap_wgrad
        sess.engine.overlap_wgrad = False
        sess.step(images)
        torch.cuda.synchronize()
        timer.enabled = True
        for _ in range(min(args.steps, 10)):
            sess.step(images)
        torch.cuda.synchronize()
        timer.enabled = False
        timer.calibrate()
        sess.engine.overlap_wgrad = overlap
    comm_report = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # per-bucket collective and EXPOSED time (how long the optimizer's stream waited for each bucket), from a few instrumented steps
        # after the timed region: makes the first multi-GPU run diagnosable (which bucket is not hidden behind backward / the update)
        sess.reducer.timing = True
        for _ in range(3):
            sess.step(images)
        torch.cuda.synchronize()
        comm_report = sess.reducer.exposed_times()
        sess.reducer.timing = False
    loss = sess.fetch()

    if rank == 0:
        total_images = world * args.batch * args.steps
        value = total_images / elapsed
        gflop = TRAIN_GFLOP_PER_IMG[args.names] * (args.size / 416.0) ** 2
        peak = BF16_DENSE_PEAK_TFLOPS if args.dtype == 'bf16' else F32_MATRIX_PEAK_TFLOPS
        out = {
            'metric': 'train_throughput', 'value': value, 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'Darknet-19 YOLOv2 %s-%d %dx%d training step (standardise+fwd+loss+bwd+Adam), batch %d per GPU (%s)'
                                   % ('VOC' if args.names == 20 else 'COCO', args.names, args.size, args.size, args.batch,
                                      'BASELINE configs[2]' if (strong and args.names == 80) else 'BASELINE configs[1]' if (args.names == 20 and args.batch == 16) else 'variant'),
                       'global_batch': world * args.batch, 'parallelism': 'dp%d' % world, 'optimizer': 'adam', 'weights': 'random-init (Xavier, seed 0)',
                       'collective': ('RCCL %s (%s, %s gradients), %d ranks, %d buckets' % ('reduce-scatter + sharded update + all-gather' if sess.shard_optimizer else 'all-reduce',
                                                                                    dist.get_backend(), grad_dtype, dist.get_world_size(), len(sess.reducer.buckets)))
                       if world > 1 else 'none (1 rank)',
                       'env_overrides': _lib_env_overrides()},       # YOLO2_* A/B switches in effect ([] = the tested defaults)
            'whole_step_tflops': value * gflop / 1e3,
            'whole_step_frac_of_mfma_peak': value * gflop / 1e3 / peak / world,
            'total_loss': loss['total_loss'],
        }
        if comm_report is not None:
            from yolo_tf_amd import ops as _ops
            out['comm'] = {'grad_dtype': grad_dtype, 'buckets': comm_report, 'exposed_ms_per_step': sum(b['exposed_ms'] or 0.0 for b in comm_report),
                           'stream_k_workgroups': _ops.get_stream_workgroups(),
                           'note': 'rank 0, last of 3 instrumented steps after the timed region; exposed = time the update stream waited for the bucket'}
        ks = timer.summary(('fwd', 'dgrad')) if timer else None       # the 3x3 launches of the implicit-GEMM kernel
        if ks:
            kf, kd, k1 = timer.summary('fwd'), timer.summary('dgrad'), timer.summary('1x1')
            raw_tflops = ks['tflops'] * ks['total_ms'] / (ks['raw_bracket_avg_ms'] * ks['launches'])      # from the uncorrected brackets (= the rocprofv3 average to 0.5 %)
            out['roofline'] = {'bound': 'mfma', 'achieved': raw_tflops, 'peak': peak, 'unit': 'TFLOP/s', 'frac': raw_tflops / peak,
                               'traffic': IGEMM_HBM_BYTES_PER_LAUNCH if (args.dtype == 'bf16' and args.batch == 16 and args.size == 416) else None,
                               'kernel': 'conv3x3_pp_kernel<...> + conv_igemm_kernel<%s, BN=128, KS=3, ...> (the 3x3 implicit-GEMM forward + data-gradient '
                                         'convolutions with > 64 filters: the "3x3 convs" of the north-star target; the ping-pong tap-fused kernel takes the '
                                         'layers on images up to 55 wide, the per-tap kernel the 104x104 stage)' % args.dtype,
                               'sustained_mfma_peak_note': 'a pure v_mfma_f32_32x32x16_bf16 loop sustains 1.9-2.1 PFLOP/s on these boxes '
                                                           '(profiles/r02_igemm_tap.md); frac is priced against the 2.5 PFLOP/s datasheet figure above',
                               'launches': ks['launches'], 'avg_launch_ms': ks['raw_bracket_avg_ms'], 'algorithmic_flop_per_launch': ks['flop_per_launch'],
                               'event_bracket': {'raw_avg_ms': ks['raw_bracket_avg_ms'], 'overhead_ms': timer.bracket_overhead_ms,
                                                 'around_empty_kernel_ms': timer.bracket_noop_ms, 'calibrated_avg_ms': ks['avg_ms'],
                                                 'calibrated_tflops': ks['tflops'], 'calibrated_frac': ks['tflops'] / peak,
                                                 'note': 'achieved / frac / avg_launch_ms come from the RAW HIP-event brackets; the calibrated figures subtract '
                                                         'the dispatch latency a bracket adds (measured on brackets around an empty kernel, its own 3.4 us excluded)'},
                               'algorithmic_bytes_per_launch': IGEMM_ALGORITHMIC_BYTES_PER_LAUNCH, 'traffic_unit': 'bytes per launch (PMC, separate passes)',
                               'traffic_source': {'file': 'profiles/dominant_kernel_pmc.json', 'measured_at_commit': PMC['measured_at_commit']},
                               'measured_over': '%d instrumented single-stream training steps run right after the timed region '
                                                '(events inside it cost 11 %% of the step)' % min(args.steps, 10),
                               'forward_launches_tflops': kf['tflops'] if kf else None, 'data_gradient_launches_tflops': kd['tflops'] if kd else None,
                               'one_by_one_launches_same_template': {'tflops': k1['tflops'], 'launches': k1['launches'], 'avg_launch_ms': k1['avg_ms']} if k1 else None}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.names, args.size)
        if world == 1 and not args.no_f32 and args.dtype == 'bf16':
            out['f32_parity_mode'] = f32_parity_mode(args)
        if world == 1 and not args.no_detect:
            del sess
            torch.cuda.empty_cache()
            out['detect'] = detect_latency(args, basedir)
        print(json.dumps(out), flush=True)
    barrier()
    if world > 1:
        dist.destroy_process_group()


def multiscale(args, rank, world, basedir, dist):
    """BASELINE configs[3]: Darknet-19 multi-scale training, input size drawn from {320, 352, ..., 608} per step (here: cycled, so
    that every run does the same work), one set of weights, buffers allocated once for 608x608.  Weak scaling over ranks."""
    from yolo_tf_amd.session import TrainSession
    from yolo_tf_amd.utils import data
    sizes = list(range(320, 609, 32))
    batch = args.batch if '--batch' in sys.argv else 8
    builder, cfg = make_builder('darknet', args.names, 416, True, basedir)
    sess = TrainSession(builder, batch, dtype=args.dtype, optimizer='adam', learning_rate=1e-6, seed=0, world_size=world,
                        bucket_mb=cfg.getfloat('mi355x', 'bucket_mb'), sizes=[(s, s) for s in sizes])
    gen = torch.Generator(device='cuda').manual_seed(1234 + rank)
    images = {}
    for sz in sizes:
        sess.set_size(sz, sz)
        images[sz] = torch.rand(batch, sz, sz, 3, device='cuda', generator=gen) * 255.0
        sess.upload_labels(data.synthetic_batch(batch, args.names, sz // 32, sz // 32, seed=4321 + rank + sz))

    def barrier():
        if world > 1:
            dist.barrier()

    def run(n):
        for i in range(n):
            sz = sizes[i % len(sizes)]
            sess.set_size(sz, sz)
            sess.step(images[sz])

    run(max(args.warmup, len(sizes)))            # every size once before the clock starts (plans, filter layouts)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    comm_report = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # per-bucket collective and EXPOSED time (how long the optimizer's stream waited for each bucket), from a few instrumented steps
        # after the timed region: makes the first multi-GPU run diagnosable (which bucket is not hidden behind backward / the update)
        sess.reducer.timing = True
        run(3)
        torch.cuda.synchronize()
        comm_report = sess.reducer.exposed_times()
        sess.reducer.timing = False
    loss = sess.fetch()
    if rank == 0:
        gflop = sum(TRAIN_GFLOP_PER_IMG[args.names] * (sizes[i % len(sizes)] / 416.0) ** 2 for i in range(args.steps)) * batch * world
        peak = BF16_DENSE_PEAK_TFLOPS if args.dtype == 'bf16' else F32_MATRIX_PEAK_TFLOPS
        print(json.dumps({
            'metric': 'train_throughput', 'value': world * batch * args.steps / elapsed, 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': max(args.warmup, len(sizes)), 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'Darknet-19 YOLOv2 multi-scale training, input size cycled over %s, batch %d per GPU (BASELINE configs[3])' % (sizes, batch),
                       'global_batch': world * batch, 'parallelism': 'dp%d' % world, 'optimizer': 'adam', 'weights': 'random-init (Xavier, seed 0)'},
            'whole_step_tflops': gflop / elapsed / 1e3, 'whole_step_frac_of_mfma_peak': gflop / elapsed / 1e3 / peak / world,
            'comm': None if comm_report is None else {'buckets': comm_report, 'exposed_ms_per_step': sum(b['exposed_ms'] or 0.0 for b in comm_report)},
            'total_loss': loss['total_loss']}), flush=True)
    barrier()
    if world > 1:
        dist.destroy_process_group()


def detect_latency(args, basedir, batch=256, iters=200):
    """BASELINE configs[4]: batch-256 416x416 detect (standardise + forward + decode + on-GPU NMS), p50/p99 latency.
    Random-init weights give conf ~ 0.5/C << 0.3, so the network's own scores leave the NMS nothing to suppress; the figure
    that includes real NMS work overwrites the decoded boxes with the stress inputs of BASELINE.md section 2 (sparse
    realistic and the reference's dense worst case, a different draw per image) between decode and NMS -- a 22 MB
    device copy that is inside the timed region."""
    from yolo_tf_amd.session import DetectSession
    b, _ = make_builder('darknet', args.names, args.size, False, basedir)
    sess = DetectSession(b, batch, dtype=args.dtype, seed=0)
    images = torch.rand(batch, args.size, args.size, 3, device='cuda') * 255.0
    cells = (args.size // 32) ** 2 * len(b.anchors)

    def pct(times):
        times = sorted(times)
        return times[len(times) // 2], times[min(len(times) - 1, int(len(times) * 0.99))]

    def run(stress):
        sess.run(images, 0, check_numerics=False)
        if stress is not None:
            sess.conf.copy_(stress[0])
            sess.xy_min.copy_(stress[1])
            sess.xy_max.copy_(stress[2])
        sess.nms(0.3, 0.4)

    out = {'batch': batch, 'iters': iters}
    modes = [('network_scores', None)]
    if cells == 845:
        for kind in ('sparse', 'dense'):
            modes.append((kind, [torch.from_numpy(a).cuda() for a in nms_stress_inputs(kind, batch, args.names, seed0=1000)]))
    for name, stress in modes:
        for _ in range(3):
            run(stress)
        torch.cuda.synchronize()
        times = []
        for _ in range(iters):
            t0 = time.perf_counter()
            run(stress)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        p50, p99 = pct(times)
        entry = {'p50_ms': p50, 'p99_ms': p99, 'img_per_s': batch / (p50 * 1e-3)}
        if stress is not None:
            # the NMS launch alone on the same inputs (HIP events around it; the scores are restored before every launch)
            ev = []
            for _ in range(10):
                sess.conf.copy_(stress[0])
                a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                sess.nms(0.3, 0.4)
                z.record()
                ev.append((a, z))
            torch.cuda.synchronize()
            nms_ms = sorted(x.elapsed_time(y) for x, y in ev)[len(ev) // 2]
            kept = int((sess.conf.max(dim=2).values > 0.3).sum().item())
            nbytes = batch * cells * (2 * args.names + 4) * 4          # scores in + out, corners in (845*(C+4)*4 B in per image)
            entry.update({'nms_only_ms': nms_ms, 'nms_us_per_image': nms_ms * 1e3 / batch, 'boxes_kept_per_image': kept / batch,
                          'nms_algorithmic_GBps': nbytes / (nms_ms * 1e-3) / 1e9,
                          'nms_bound': 'latency / LDS (O(C*N^2) IoU tests on an 81 KB per-image payload): HBM roofline fraction %.4f'
                                       % (nbytes / (nms_ms * 1e-3) / 8e12)})
        out[name] = entry
    # the headline detect latency is the one that contains NMS work: the dense worst case when available
    head = out.get('dense', out['network_scores'])
    out.update({'p50_ms': head['p50_ms'], 'p99_ms': head['p99_ms'], 'img_per_s': head['img_per_s'],
                'headline': 'dense NMS stress input (every box a candidate)' if 'dense' in out else 'network scores'})
    return out


if __name__ == '__main__':
    main()

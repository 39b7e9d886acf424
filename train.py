#!/usr/bin/env python
"""train.py -- YOLO/YOLOv2 training on MI355X with the reference CLI (ruiminshen/yolo-tf train.py:148-167):

    python train.py -c config.ini config/yolo2/darknet-20.ini -b 16 -o adam -lr 1e-4 -s 1000
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py -c ... -b 8

Same flags (-c -t -e -p -s -d -b -o -n -g -lr --seed --summary_secs --save_secs --level --master
--task); same logdir/cachedir layout, auto-resume from the latest checkpoint in logdir, `-d` wipes
it, `-t/-e` transfers variables.  One process per GPU; `-b` is the per-GPU batch.  The reference's
TFRecord/JPEG input pipeline is out of the hot-path scope: batches come from
``--data synthetic`` (seeded generator, default) or ``--data file.npz``: either pre-made labels (images uint8
[N,H,W,3] + the six tensors of utils.data.transform_labels) or decoded images with raw boxes (images,
objects_class [total], objects_coord [total,4] pixels, objects_first [N+1]); the latter goes through the on-device
input pipeline (utils/augment.py: crop / resize / colour augmentation per config.ini [data_augmentation_*], labels
built on the GPU) -- the counterpart of the reference's load_image_labels after JPEG decode.  ``--data cache`` reads the
reference's own TFRecord cache (<cachedir>/<profile>.tfrecord for every ``-p`` profile, utils/tfrecord.py) into HBM."""
import argparse
import configparser
import logging
import os
import re
import shutil
import time

import numpy as np
import torch

from yolo_tf_amd import checkpoint, tf_checkpoint, utils
from yolo_tf_amd.utils import events
from yolo_tf_amd.parallel import agree, init_distributed, sync_replicas
from yolo_tf_amd.session import TrainSession
from yolo_tf_amd.utils import data as udata


def make_args():
    parser = argparse.ArgumentParser()
    parser.add_argument('-c', '--config', nargs='+', default=['config.ini'], help='config file')
    parser.add_argument('-t', '--transfer', help='transferring model from a checkpoint file')
    parser.add_argument('-e', '--exclude', nargs='+', help='exclude variables while transferring')
    parser.add_argument('-p', '--profile', nargs='+', default=['train', 'val'])
    parser.add_argument('-s', '--steps', type=int, default=None, help='max number of steps')
    parser.add_argument('-d', '--delete', action='store_true', help='delete logdir')
    parser.add_argument('-b', '--batch_size', default=8, type=int, help='batch size (per GPU)')
    parser.add_argument('-o', '--optimizer', default='adam')
    parser.add_argument('-n', '--logname', default=time.strftime('%Y-%m-%d_%H-%M-%S'), help='run name')
    parser.add_argument('-g', '--gradient_clip', default=0, type=float, help='gradient clip')
    parser.add_argument('-lr', '--learning_rate', default=1e-6, type=float, help='learning rate')
    parser.add_argument('--seed', type=int, default=None)
    parser.add_argument('--summary_secs', default=30, type=int, help='seconds between logged summaries')
    parser.add_argument('--save_secs', default=600, type=int, help='seconds to save model')
    parser.add_argument('--level', help='logging level')
    parser.add_argument('--master', default='', help='accepted for compatibility (rendezvous comes from torch.distributed.run)')
    parser.add_argument('--task', type=int, default=0, help='accepted for compatibility (rank comes from RANK)')
    parser.add_argument('--data', default='cache', help="'cache' (default, as the reference's train.py:98-100: the TFRecord cache of the -p profiles under cachedir), "
                        "'synthetic' (random images and labels: benchmarking / smoke runs) or a .npz file")
    parser.add_argument('--dtype', default=None, choices=['bf16', 'f32'], help='overrides [mi355x] dtype')
    parser.add_argument('--ckpt_format', default='npz', choices=['npz', 'tf'], help="checkpoint container: 'npz' (every optimizer) or 'tf' (TensorFlow V2 bundle, as the reference's tf.train.Saver writes)")
    return parser.parse_args()


class SyntheticData(object):
    def __init__(self, batch, classes, width, height, cell_width, cell_height, seed):
        self.args = (batch, classes, cell_width, cell_height)
        self.gen = torch.Generator(device='cuda').manual_seed(seed)
        self.shape = (batch, height, width, 3)
        self.seed = seed
        self.i = 0

    def next(self):
        images = torch.rand(*self.shape, device='cuda', generator=self.gen) * 255.0
        labels = udata.synthetic_batch(*self.args, seed=self.seed * 1000003 + self.i)
        self.i += 1
        return images, labels


class NpzData(object):
    def __init__(self, path, batch, seed, rank, world):
        z = np.load(path)
        self.images = z['images']
        self.labels = [z[k] for k in udata.LABEL_KEYS]
        self.batch = batch
        self.rng = np.random.RandomState(seed)
        self.rank, self.world = rank, world

    def next(self):
        idx = self.rng.randint(0, len(self.images), self.batch * self.world)[self.rank::self.world]
        images = torch.from_numpy(self.images[idx].astype(np.float32)).cuda()
        return images, tuple(l[idx] for l in self.labels)


class DeviceAugmentedData(object):
    """Decoded images + raw boxes resident in HBM -> augmented batch and label tensors, all on the device."""

    def __init__(self, z, batch, width, height, classes, cell_width, cell_height, config, seed, rank, world, session):
        from yolo_tf_amd.utils.augment import DeviceInputPipeline
        if isinstance(z, tuple):          # (images, objects) already decoded, e.g. from the reference's TFRecord cache
            images, objects = z
        else:
            images = list(z['images'])
            first = z['objects_first']
            objects = [(z['objects_class'][first[i]:first[i + 1]], z['objects_coord'][first[i]:first[i + 1]]) for i in range(len(images))]
        self.pipe = DeviceInputPipeline(images, objects, batch, width, height, classes, cell_width, cell_height, config=config,
                                        seed=seed, rank=rank, world=world)
        self.session = session

    def next(self):
        return self.pipe.next(self.session), None          # labels were written into the session's buffers in place


def main():
    rank, local_rank, world = init_distributed()
    torch.cuda.set_device(local_rank)
    model = config.get('config', 'model')
    logdir = utils.get_logdir(config)
    if args.delete and rank == 0:
        logging.warning('delete logging directory: ' + logdir)
        shutil.rmtree(logdir, ignore_errors=True)
    if world > 1:
        torch.distributed.barrier()          # nobody looks for checkpoints while rank 0 is still deleting them
    utils.ensure_names(config)
    width = config.getint(model, 'width')
    height = config.getint(model, 'height')
    cell_width, cell_height = utils.calc_cell_width_height(config, width, height)
    logging.warning('(width, height)=(%d, %d), (cell_width, cell_height)=(%d, %d)' % (width, height, cell_width, cell_height))
    yolo = __import__('yolo_tf_amd.model.' + model, fromlist=['Builder'])
    builder = yolo.Builder(args, config)
    builder(None, training=True)
    builder.create_objectives()
    dtype = args.dtype or (config.get('mi355x', 'dtype') if config.has_option('mi355x', 'dtype') else 'bf16')
    seed = args.seed if args.seed is not None else 0
    session = TrainSession(builder, args.batch_size, dtype=dtype, optimizer=args.optimizer, learning_rate=args.learning_rate,
                           gradient_clip=args.gradient_clip, config=config, seed=seed, world_size=world,
                           bucket_mb=config.getfloat('mi355x', 'bucket_mb') if config.has_option('mi355x', 'bucket_mb') else 64.0,
                           grad_dtype=config.get('mi355x', 'grad_dtype') if config.has_option('mi355x', 'grad_dtype') else 'f32',
                           sync_bn=config.getboolean('mi355x', 'sync_bn') if config.has_option('mi355x', 'sync_bn') else False,
                           shard_optimizer=config.getboolean('mi355x', 'shard_optimizer') if config.has_option('mi355x', 'shard_optimizer') else False)
    logging.warning('optimizer=%s, dtype=%s, world=%d, parameters=%d' % (args.optimizer, dtype, world, session.engine.n_params))
    # rank 0 alone chooses and reads the checkpoint; the others receive parameters, statistics, optimizer slots and
    # global_step from it (same seed -> same initial weights anyway, but a restore must not depend on what each rank sees)
    # Both containers may sit in one logdir (a run resumed with the other --ckpt_format, or a logdir the reference wrote): the one
    # with the HIGHER global step wins, so a restart never falls back to an older state.  A failure on rank 0 (corrupt file, shape
    # mismatch) is agreed on before anyone enters the broadcast of sync_replicas -- the other ranks would wait there forever.
    error = None
    if rank == 0:
        try:
            latest = checkpoint.latest_checkpoint(logdir)
            latest_tf = tf_checkpoint.latest_checkpoint(logdir)            # a logdir written by the reference (or --ckpt_format tf)
            step_npz = int(re.search(r'model\.ckpt-(\d+)\.npz$', latest).group(1)) if latest else -1
            if latest and step_npz >= tf_checkpoint.checkpoint_step(latest_tf):
                step = checkpoint.restore(latest, session)
                logging.warning('resuming from %s (global_step=%d)' % (latest, step))
            elif latest_tf:
                step = tf_checkpoint.restore(latest_tf, session)
                logging.warning('resuming from TensorFlow checkpoint %s (global_step=%d)' % (latest_tf, step))
            elif args.transfer:
                path = os.path.expanduser(os.path.expandvars(args.transfer))
                logging.warning('transferring from ' + path)
                if os.path.exists(path + '.index'):                     # a TensorFlow checkpoint prefix, like the reference's -t argument
                    tf_checkpoint.restore(path, session, exclude=args.exclude, variables_only=True)
                else:
                    checkpoint.restore(path, session, exclude=args.exclude, variables_only=True)
        except Exception as e:       # noqa: BLE001  (re-raised below, on every rank)
            error = e
    (failed,) = agree([error is not None], session.engine.device)
    if failed:
        raise error if error is not None else RuntimeError('rank 0 could not restore the checkpoint (see its log)')
    sync_replicas(session)
    logging.warning('global_step=%d, learning_rate=%g' % (session.global_step, session.lr_fn(session.global_step)))
    if args.data == 'synthetic':
        data = SyntheticData(args.batch_size, len(builder.names), width, height, cell_width, cell_height, seed * world + rank + 1)
    elif args.data == 'cache':
        # the reference's own dataset cache: <cachedir>/<profile>.tfrecord written by its cache.py (train.py:98-100 there)
        from yolo_tf_amd.utils import tfrecord
        cachedir = utils.get_cachedir(config)
        paths = [os.path.join(cachedir, profile + '.tfrecord') for profile in args.profile]
        logging.warning('loading ' + ', '.join(paths))
        data = DeviceAugmentedData(tfrecord.load_dataset(paths), args.batch_size, width, height, len(builder.names), cell_width, cell_height,
                                   config, seed, rank, world, session)
    elif 'objects_coord' in np.load(args.data, allow_pickle=True).files:
        data = DeviceAugmentedData(np.load(args.data, allow_pickle=True), args.batch_size, width, height, len(builder.names), cell_width, cell_height,
                                   config, seed, rank, world, session)
    else:
        data = NpzData(args.data, args.batch_size, seed, rank, world)
    # the reference's summary_writer (train.py:141-145): scalar events under <logdir>/<logname>
    writer = events.FileWriter(os.path.join(logdir, args.logname)) if rank == 0 else None
    save = (lambda: tf_checkpoint.save(logdir, session)) if args.ckpt_format == 'tf' else (lambda: checkpoint.save(logdir, session))
    last_summary = last_save = t_rate = time.time()
    n_rate = 0
    sync_every = 1 if world == 1 else 20     # data parallel: decisions only one rank can make are agreed on every 20 steps
    device = session.engine.device
    while args.steps is None or session.global_step < args.steps:
        images, labels = data.next()
        session.step(images, labels)
        n_rate += 1
        last = args.steps is not None and session.global_step >= args.steps
        if not (last or session.global_step % sync_every == 0):
            continue
        now = time.time()
        # every rank takes the same branches: rank 0's clock decides, any rank's non-finite loss stops all of them
        # (a rank that raised alone would leave the others waiting in the next all-reduce)
        want_summary = rank == 0 and (now - last_summary >= args.summary_secs or last)
        want_save = rank == 0 and now - last_save >= args.save_secs
        want_summary, want_save = agree([want_summary, want_save], device)
        # device-detected failures (a stream-K tile owner that gave up waiting: include/yolo2_hip.h yolo2_check_async_errors).  The per-step
        # snapshot is polled without a synchronisation; before anything is summarised or written the synchronising check runs.  Either way the
        # error goes through agree() like shard_error below: one rank raising alone would leave its peers waiting in the next all-reduce, and
        # a checkpoint must not be written from parameters that garbage gradients have updated.
        dev_error = session.device_error() if (want_summary or want_save or last or session.async_error_pending()) else None
        (bad_device,) = agree([dev_error is not None], device)
        if bad_device:
            raise dev_error if dev_error is not None else RuntimeError('another rank reported a device-side failure (stream-K hand-off gave up)')
        if want_summary:
            s = session.fetch()
            shard_error = None
            if hasattr(data, 'pipe'):
                try:
                    data.pipe.check()    # objects outside the grid / bad class ids / negative extents: the reference raises there
                except Exception as e:   # noqa: BLE001  (a shard-local error: agreed on, then raised by every rank)
                    shard_error = e
            (bad_shard,) = agree([shard_error is not None], device)
            if bad_shard:
                raise shard_error if shard_error is not None else RuntimeError('another rank found invalid objects in its data shard')
            rate = n_rate * args.batch_size * world / (time.time() - t_rate)
            if rank == 0:
                writer.add_training_summary(session.global_step, s)
                writer.flush()
                logging.warning('step %d: total_loss=%.6f iou_best=%.6f iou_normal=%.6f coords=%.6f prob=%.6f (%.1f img/s)'
                                % (session.global_step, s['total_loss'], s['iou_best'], s['iou_normal'], s['coords'], s['prob'], rate))
            (bad,) = agree([not np.isfinite(s['total_loss'])], device)
            if bad:
                raise FloatingPointError('total_loss is not finite (on at least one rank)')
            last_summary, t_rate, n_rate = now, time.time(), 0
        if want_save:
            session.gather_optimizer_state()     # (optimizer sharding: every rank joins the all-gather of the slot shards; a no-op otherwise)
            if rank == 0:
                logging.warning('saved ' + save())
            last_save = now
    session.gather_optimizer_state()
    if rank == 0:
        logging.warning('saved ' + save())
        writer.close()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    args = make_args()
    config = configparser.ConfigParser()
    utils.load_config(config, args.config)
    logging.basicConfig(format='%(asctime)s %(message)s')
    if args.level:
        logging.getLogger().setLevel(args.level.upper())
    main()

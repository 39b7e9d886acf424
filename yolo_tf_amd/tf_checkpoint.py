"""TensorFlow checkpoint (V2 "tensor bundle") reader and writer without TensorFlow.

The reference saves and restores through ``tf.train.Saver`` (slim.learning.train, train.py:141-145;
``slim.assign_from_checkpoint(_fn)`` train.py:130-136, detect.py:104-106): ``model.ckpt-<step>.index`` +
``model.ckpt-<step>.data-00000-of-00001`` in the logdir, variables named by their TF scopes
(``yolo2_darknet/conv3/BatchNorm/gamma`` ...; this package uses the same names), plus ``global_step`` and the optimizer slots
(``<var>/Adam``, ``<var>/Adam_1``, ``optimizer/beta1_power`` ...).  This module reads such a pair into ``{name: ndarray}`` and writes
one from it, so existing ``model.ckpt-*`` files can be restored into ``Engine.set_variables`` and weights trained here can go back.

Format (tensorflow/core/util/tensor_bundle, TF >= 0.12): the ``.index`` file is an SSTable in LevelDB's table format -- data blocks
of prefix-compressed (key, value) entries with restart points, each block followed by a 1-byte compression type (0: TF writes
bundles uncompressed; snappy blocks are rejected) and a masked CRC32C; an (empty) metaindex block; an index block mapping
separator keys to block handles; a 48-byte footer ending in the magic 0xdb4775248b80fb57.  Key "" holds ``BundleHeaderProto``
{num_shards=1, endianness=0, version}; every other key is a tensor name whose value is ``BundleEntryProto`` {dtype=1, shape=2,
shard_id=3, offset=4, size=5, crc32c=6 (masked CRC32C of the tensor bytes)}.  The data shard is the tensors' raw little-endian
bytes back to back.

PARITY UNPINNED: no TensorFlow and no TF-written checkpoint exists in this image (nor in /root/reference), so the writer and the
reader are checked against each other, against the published LevelDB table layout in an independent minimal parser
(tests/test_tf_formats_cpu.py) and for checksum / corruption handling -- not against bytes TensorFlow produced."""
import os
import struct

import numpy as np

from .utils import tfrecord

MAGIC = 0xdb4775248b80fb57
DT = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 4: np.uint8, 10: np.bool_, 6: np.int8, 5: np.int16}
DT_OF = {np.dtype(v): k for k, v in DT.items()}


def _mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    return tfrecord._varint(n)


def _read_varint(b, i):
    return tfrecord._read_varint(b, i)


# ---------------------------------------------------------------- protos (hand-rolled: three small messages)
def _entry(dtype, shape, offset, size, crc):
    dims = b''.join(tfrecord._field(2, _varint(1 << 3) + _varint(int(d))) for d in shape)
    out = _varint(1 << 3) + _varint(dtype) + tfrecord._field(2, dims)
    if offset:
        out += _varint(4 << 3) + _varint(offset)
    out += _varint(5 << 3) + _varint(size) + _varint((6 << 3) | 5) + struct.pack('<I', crc)
    return out


def _parse_entry(b):
    e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None, 'slices': False}
    i = 0
    while i < len(b):
        key, i = _read_varint(b, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _read_varint(b, i)
            if f == 1:
                e['dtype'] = v
            elif f == 3:
                e['shard_id'] = v
            elif f == 4:
                e['offset'] = v
            elif f == 5:
                e['size'] = v
        elif wt == 5:
            if f == 6:
                e['crc32c'] = struct.unpack('<I', b[i:i + 4])[0]
            i += 4
        elif wt == 2:
            n, i = _read_varint(b, i)
            sub = b[i:i + n]
            i += n
            if f == 2:                      # TensorShapeProto: repeated Dim dim = 2 { int64 size = 1 }
                j = 0
                while j < len(sub):
                    k2, j = _read_varint(sub, j)
                    if k2 & 7 == 2:
                        m, j = _read_varint(sub, j)
                        dim = sub[j:j + m]
                        j += m
                        if k2 >> 3 == 2:
                            size, q = 0, 0
                            while q < len(dim):
                                k3, q = _read_varint(dim, q)
                                if k3 & 7 == 0:
                                    val, q = _read_varint(dim, q)
                                    if k3 >> 3 == 1:
                                        size = val
                                elif k3 & 7 == 2:
                                    m3, q = _read_varint(dim, q)
                                    q += m3
                            e['shape'].append(size)
                    elif k2 & 7 == 0:
                        _, j = _read_varint(sub, j)
            elif f == 7:
                e['slices'] = True
        elif wt == 1:
            i += 8
    return e


# ---------------------------------------------------------------- LevelDB table
def _block(entries, restart_interval=16):
    out, restarts, last = bytearray(), [], b''
    for n, (k, v) in enumerate(entries):
        shared = 0
        if n % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        out += _varint(shared) + _varint(len(k) - shared) + _varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts))
    return bytes(out)


def _parse_block(b):
    (nrestart,) = struct.unpack('<I', b[-4:])
    end = len(b) - 4 - 4 * nrestart
    i, key = 0, b''
    while i < end:
        shared, i = _read_varint(b, i)
        non_shared, i = _read_varint(b, i)
        vlen, i = _read_varint(b, i)
        key = key[:shared] + b[i:i + non_shared]
        i += non_shared
        yield key, b[i:i + vlen]
        i += vlen


def _handle(offset, size):
    return _varint(offset) + _varint(size)


def write(prefix, tensors, block_bytes=4096):
    """Writes ``<prefix>.index`` and ``<prefix>.data-00000-of-00001`` from ``{name: array}`` (float32 / int32 / int64 / ...)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    names = sorted(tensors, key=lambda s: s.encode())
    entries = [(b'', _varint(1 << 3) + _varint(1) + tfrecord._field(3, _varint(1 << 3) + _varint(1)))]   # header: num_shards 1, little endian, producer 1
    offset = 0
    with open(prefix + '.data-00000-of-00001.tmp', 'wb') as f:
        for name in names:
            a = np.asarray(tensors[name])
            a = a if a.flags.c_contiguous else np.ascontiguousarray(a)
            if a.dtype not in DT_OF:
                raise TypeError('%s: dtype %s has no TF checkpoint counterpart here' % (name, a.dtype))
            raw = a.tobytes()
            f.write(raw)
            entries.append((name.encode(), _entry(DT_OF[a.dtype], a.shape, offset, len(raw), _mask(tfrecord.crc32c(raw)))))
            offset += len(raw)
    # data blocks
    out = bytearray()
    index = []
    cur, cur_bytes = [], 0

    def flush():
        nonlocal cur, cur_bytes
        if not cur:
            return
        blk = _block(cur)
        index.append((cur[-1][0], len(out), len(blk)))
        out.extend(blk + b'\x00' + struct.pack('<I', _mask(tfrecord.crc32c(blk + b'\x00'))))
        cur, cur_bytes = [], 0

    for k, v in entries:
        cur.append((k, v))
        cur_bytes += len(k) + len(v)
        if cur_bytes >= block_bytes:
            flush()
    flush()
    meta = _block([])
    meta_off = len(out)
    out.extend(meta + b'\x00' + struct.pack('<I', _mask(tfrecord.crc32c(meta + b'\x00'))))
    idx = _block([(k, _handle(o, n)) for k, o, n in index], restart_interval=1)
    idx_off = len(out)
    out.extend(idx + b'\x00' + struct.pack('<I', _mask(tfrecord.crc32c(idx + b'\x00'))))
    footer = _handle(meta_off, len(meta)) + _handle(idx_off, len(idx))
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', MAGIC)
    out.extend(footer)
    with open(prefix + '.index.tmp', 'wb') as f:
        f.write(out)
    os.replace(prefix + '.data-00000-of-00001.tmp', prefix + '.data-00000-of-00001')
    os.replace(prefix + '.index.tmp', prefix + '.index')
    return prefix


def _read_block(buf, offset, size, verify):
    blk, trailer = buf[offset:offset + size], buf[offset + size:offset + size + 5]
    if len(blk) != size or len(trailer) != 5:
        raise IOError('truncated table block')
    if trailer[0] != 0:
        raise IOError('compressed table block (type %d): TF writes tensor bundles uncompressed; not supported' % trailer[0])
    if verify and struct.unpack('<I', trailer[1:])[0] != _mask(tfrecord.crc32c(blk + trailer[:1])):
        raise IOError('corrupted table block (checksum)')
    return blk


def read_index(prefix, verify=True):
    """-> {name: entry dict} (and the header under key '')."""
    with open(prefix + '.index', 'rb') as f:
        buf = f.read()
    if len(buf) < 48 or struct.unpack('<Q', buf[-8:])[0] != MAGIC:
        raise IOError('%s.index is not a TF checkpoint index (bad magic)' % prefix)
    footer = buf[-48:]
    i = 0
    _, i = _read_varint(footer, i)          # metaindex handle (unused)
    _, i = _read_varint(footer, i)
    idx_off, i = _read_varint(footer, i)
    idx_size, i = _read_varint(footer, i)
    out = {}
    for _, h in _parse_block(_read_block(buf, idx_off, idx_size, verify)):
        o, j = _read_varint(h, 0)
        n, j = _read_varint(h, j)
        for k, v in _parse_block(_read_block(buf, o, n, verify)):
            out[k.decode()] = v if k == b'' else _parse_entry(v)
    return out


def read(prefix, names=None, verify=True):
    """Reads a checkpoint into ``{name: ndarray}``; ``names`` restricts it (default: every tensor).  Checksums are verified."""
    index = read_index(prefix, verify)
    header = index.pop('', None)
    shards = {}
    out = {}
    for name, e in index.items():
        if names is not None and name not in names:
            continue
        if e['slices']:
            raise IOError('%s: partitioned variables are not supported' % name)
        if e['dtype'] not in DT:
            raise IOError('%s: unsupported dtype %d' % (name, e['dtype']))
        if e['shard_id'] not in shards:
            nsh = 1
            if header:
                j = 0
                while j < len(header):
                    k, j = _read_varint(header, j)
                    if k & 7 == 0:
                        v, j = _read_varint(header, j)
                        if k >> 3 == 1:
                            nsh = v
                    elif k & 7 == 2:
                        m, j = _read_varint(header, j)
                        j += m
            shards[e['shard_id']] = np.memmap('%s.data-%05d-of-%05d' % (prefix, e['shard_id'], nsh), dtype=np.uint8, mode='r')
        raw = bytes(shards[e['shard_id']][e['offset']:e['offset'] + e['size']])
        if len(raw) != e['size']:
            raise IOError('%s: data shard is truncated' % name)
        if verify and e['crc32c'] is not None and _mask(tfrecord.crc32c(raw)) != e['crc32c']:
            raise IOError('%s: tensor bytes are corrupted (checksum)' % name)
        out[name] = np.frombuffer(raw, DT[e['dtype']]).reshape(e['shape']).copy()
    return out


def latest_checkpoint(logdir):
    """tf.train.latest_checkpoint: the prefix named by the ``checkpoint`` state file, else the highest model.ckpt-<step>.index."""
    import glob
    import re
    state = os.path.join(logdir, 'checkpoint')
    if os.path.exists(state):
        for line in open(state):
            m = re.match(r'\s*model_checkpoint_path:\s*"(.*)"', line)
            if m:
                p = m.group(1)
                p = p if os.path.isabs(p) else os.path.join(logdir, p)
                if os.path.exists(p + '.index'):
                    return p
    best, step = None, -1
    for p in glob.glob(os.path.join(logdir, 'model.ckpt-*.index')):
        m = re.search(r'model\.ckpt-(\d+)\.index$', p)
        if m and int(m.group(1)) > step:
            best, step = p[:-len('.index')], int(m.group(1))
    return best


def checkpoint_step(prefix):
    """global step in a checkpoint prefix's name (``model.ckpt-<step>``), -1 when it has none."""
    import re
    m = re.search(r'model\.ckpt-(\d+)$', prefix or '')
    return int(m.group(1)) if m else -1


def restore(prefix, session=None, engine=None, exclude=None, variables_only=False):
    """Variables of a TF checkpoint into the engine by name (``slim.assign_from_checkpoint_fn(model_path, tf.global_variables())``,
    detect.py:104-106): every graph variable found in the file is assigned; ``global_step`` and Adam's ``<var>/Adam``,
    ``<var>/Adam_1`` slots go into a TrainSession when one is given.  ``variables_only`` (the reference's ``-t ckpt -e scope...``
    transfer, train.py:114,130-136): no optimizer slots, and ``global_step`` is taken unless ``exclude`` names it -- as
    slim.get_variables_to_restore(exclude=...) does, so exponential_decay continues from the donor's step (checkpoint.restore
    behaves the same).  Returns global_step (0 when absent)."""
    engine = engine if engine is not None else session.engine
    index = read_index(prefix)
    index.pop('', None)
    wanted = [v.name for v in engine.graph.variables.values() if v.name in index and not (exclude and any(v.name.startswith(s) for s in exclude))]
    extra = ['global_step'] if 'global_step' in index else []
    slots = []
    if session is not None and session.optimizer.name == 'adam' and not variables_only:
        slots = [n + sfx for n in wanted for sfx in ('/Adam', '/Adam_1') if n + sfx in index]
    values = read(prefix, set(wanted + extra + slots))
    engine.set_variables({k: values[k] for k in wanted}, strict=False)
    step = int(values['global_step']) if 'global_step' in values else 0
    if session is not None:
        if not (variables_only and exclude and any('global_step'.startswith(s) for s in exclude)):
            session.global_step = step
        if slots:
            import torch
            for n in wanted:
                if n in engine.param_offsets:
                    o, sz = engine.param_offsets[n]
                    for slot, sfx in zip(session.optimizer.slots, ('/Adam', '/Adam_1')):
                        if n + sfx in values:
                            slot[o:o + sz].copy_(torch.from_numpy(np.ascontiguousarray(values[n + sfx], np.float32).reshape(-1)))
    return step


def save(logdir, session, step=None, keep=5):
    """Writes ``<logdir>/model.ckpt-<step>`` (+ the ``checkpoint`` state file) with the reference's variable names, global_step and
    the Adam slots, i.e. what its tf.train.Saver would hold for this model.  Keeps the ``keep`` most recent checkpoints
    ([TF-sem] tf.train.Saver max_to_keep=5: each is ~0.8 GB with Adam slots) and lists them in all_model_checkpoint_paths."""
    if not getattr(session, 'optimizer_state_complete', True):
        raise RuntimeError('optimizer sharding: this rank holds the optimizer slots of its own shards only; call '
                           'session.gather_optimizer_state() on EVERY rank before saving (train.py does)')
    e = session.engine
    step = session.global_step if step is None else step
    tensors = dict(e.get_variables())
    tensors['global_step'] = np.int64(step)
    if session.optimizer.name == 'adam':
        for slot, sfx in zip(session.optimizer.slots, ('/Adam', '/Adam_1')):
            host = slot.cpu().numpy()
            for v in e.graph.trainable():
                o, sz = e.param_offsets[v.name]
                tensors[v.name + sfx] = host[o:o + sz].reshape(v.shape)
    prefix = os.path.join(logdir, 'model.ckpt-%d' % step)
    write(prefix, tensors)
    import glob
    found = sorted((checkpoint_step(q[:-len('.index')]), q[:-len('.index')]) for q in glob.glob(os.path.join(logdir, 'model.ckpt-*.index')))
    found = [f for f in found if f[0] >= 0]
    if keep and keep > 0:
        kept = [f for f in found if f[1] == prefix or f in found[-keep:]]
        for f in found:
            if f not in kept:
                for path in glob.glob(f[1] + '.index') + glob.glob(f[1] + '.data-*'):
                    os.remove(path)
        found = kept
    with open(os.path.join(logdir, 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "model.ckpt-%d"\n' % step)
        for st, _ in found:
            f.write('all_model_checkpoint_paths: "model.ckpt-%d"\n' % st)
    return prefix

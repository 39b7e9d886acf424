"""the reference's TensorFlow artefact formats read and written without TensorFlow -- TensorBoard event files
(train.py:141-145 summary_writer, config.ini:63 scalars) and V2 checkpoints (tf.train.Saver via slim, train.py:130-145,
detect.py:104-106).  Proto encodings are checked against google.protobuf's serializer on the published schemas (an independent
implementation of the wire format); the LevelDB table layout against a second, minimal parser written here from the format
description; checksums against corruption.  No TF-written file exists in this image: see yolo_tf_amd/C.py's header."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yolo_tf_amd import tf_checkpoint as C        # noqa: E402
from yolo_tf_amd.utils import events as E          # noqa: E402
from yolo_tf_amd.utils import tfrecord as T        # noqa: E402


def _schemas():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    f = descriptor_pb2.FileDescriptorProto(name='tf_formats_schema.proto', package='tfy', syntax='proto3')
    FD = descriptor_pb2.FieldDescriptorProto

    def msg(name):
        m = f.message_type.add()
        m.name = name
        return m
    # tensorflow/core/framework/summary.proto + core/util/event.proto (the fields used)
    m = msg('Value')
    m.field.add(name='tag', number=1, label=FD.LABEL_OPTIONAL, type=FD.TYPE_STRING)
    m.field.add(name='simple_value', number=2, label=FD.LABEL_OPTIONAL, type=FD.TYPE_FLOAT)
    m = msg('Summary'); m.field.add(name='value', number=1, label=FD.LABEL_REPEATED, type=FD.TYPE_MESSAGE, type_name='.tfy.Value')
    m = msg('Event')
    m.oneof_decl.add(name='what')
    m.field.add(name='wall_time', number=1, label=FD.LABEL_OPTIONAL, type=FD.TYPE_DOUBLE)
    m.field.add(name='step', number=2, label=FD.LABEL_OPTIONAL, type=FD.TYPE_INT64)
    m.field.add(name='file_version', number=3, label=FD.LABEL_OPTIONAL, type=FD.TYPE_STRING, oneof_index=0)
    m.field.add(name='summary', number=5, label=FD.LABEL_OPTIONAL, type=FD.TYPE_MESSAGE, type_name='.tfy.Summary', oneof_index=0)
    # tensorflow/core/protobuf/tensor_bundle.proto + framework/tensor_shape.proto
    m = msg('Dim'); m.field.add(name='size', number=1, label=FD.LABEL_OPTIONAL, type=FD.TYPE_INT64)
    m = msg('TensorShapeProto'); m.field.add(name='dim', number=2, label=FD.LABEL_REPEATED, type=FD.TYPE_MESSAGE, type_name='.tfy.Dim')
    m = msg('BundleEntryProto')
    m.field.add(name='dtype', number=1, label=FD.LABEL_OPTIONAL, type=FD.TYPE_INT32)
    m.field.add(name='shape', number=2, label=FD.LABEL_OPTIONAL, type=FD.TYPE_MESSAGE, type_name='.tfy.TensorShapeProto')
    m.field.add(name='shard_id', number=3, label=FD.LABEL_OPTIONAL, type=FD.TYPE_INT32)
    m.field.add(name='offset', number=4, label=FD.LABEL_OPTIONAL, type=FD.TYPE_INT64)
    m.field.add(name='size', number=5, label=FD.LABEL_OPTIONAL, type=FD.TYPE_INT64)
    m.field.add(name='crc32c', number=6, label=FD.LABEL_OPTIONAL, type=FD.TYPE_FIXED32)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName('tfy.' + n))
    return get('Event'), get('BundleEntryProto')


def test_event_encoding_matches_protobuf_and_round_trips(tmp_path):
    pytest.importorskip('google.protobuf')
    Event, _ = _schemas()
    ev = Event(wall_time=1234.5, step=42)
    for tag, v in zip(E.SCALAR_TAGS, (1.25, 0.5, 0.125, 3.0, 7.5)):
        ev.summary.value.add(tag=tag, simple_value=v)
    mine = E.encode_event(1234.5, step=42, scalars=list(zip(E.SCALAR_TAGS, (1.25, 0.5, 0.125, 3.0, 7.5))))
    assert mine == ev.SerializeToString(deterministic=True)
    assert E.encode_event(9.0, file_version='brain.Event:2') == Event(wall_time=9.0, file_version='brain.Event:2').SerializeToString()
    w = E.FileWriter(str(tmp_path / 'run'))
    assert os.path.basename(w.path).startswith('events.out.tfevents.')
    w.add_training_summary(7, {'total_loss': 1.5, 'iou_best': .25, 'iou_normal': .5, 'coords': .75, 'prob': 1.0})
    w.add_scalars(8, {'total_loss': 2.0})
    w.close()
    got = E.read_events(w.path)
    assert got[0]['file_version'] == 'brain.Event:2' and got[0]['step'] == 0
    assert got[1]['step'] == 7 and got[1]['scalars'] == list(zip(E.SCALAR_TAGS, (1.5, .25, .5, .75, 1.0)))
    assert got[2]['scalars'] == [('total_loss', 2.0)]
    # protobuf parses every record the same way
    recs = list(T.read_records(w.path))
    p = Event.FromString(recs[1])
    assert p.step == 7 and [v.tag for v in p.summary.value] == list(E.SCALAR_TAGS) and p.summary.value[0].simple_value == 1.5
    raw = bytearray(open(w.path, 'rb').read())
    raw[-6] ^= 0x40
    open(w.path, 'wb').write(raw)
    with pytest.raises(IOError):
        E.read_events(w.path)


def _tensors(rng):
    t = {'global_step': np.int64(12345), 'yolo2_darknet/conv0/weights': rng.randn(3, 3, 3, 32).astype(np.float32),
         'yolo2_darknet/conv0/BatchNorm/gamma': rng.rand(32).astype(np.float32), 'optimizer/beta1_power': np.float32(0.9),
         'yolo2_darknet/conv20/weights': rng.randn(3, 3, 48, 64).astype(np.float32), 'ints': np.arange(7, dtype=np.int32)}
    for i in range(300):            # enough keys for several data blocks and restart points
        t['yolo2_darknet/conv%d/BatchNorm/moving_mean' % i] = rng.randn(5).astype(np.float32)
    return t


def test_checkpoint_round_trip_and_entry_encoding(tmp_path):
    pytest.importorskip('google.protobuf')
    _, Entry = _schemas()
    rng = np.random.RandomState(0)
    t = _tensors(rng)
    prefix = str(tmp_path / 'model.ckpt-12345')
    C.write(prefix, t)
    assert os.path.exists(prefix + '.index') and os.path.exists(prefix + '.data-00000-of-00001')
    got = C.read(prefix)
    assert set(got) == set(t)
    for k in t:
        assert got[k].dtype == np.asarray(t[k]).dtype and got[k].shape == np.shape(t[k]) and np.array_equal(got[k], t[k]), k
    only = C.read(prefix, names={'global_step'})
    assert list(only) == ['global_step'] and int(only['global_step']) == 12345
    # BundleEntryProto bytes == protobuf's encoding of the same message
    idx = C.read_index(prefix)
    e = idx['yolo2_darknet/conv0/weights']
    pb = Entry(dtype=1, offset=e['offset'], size=e['size'], crc32c=e['crc32c'])
    for d in (3, 3, 3, 32):
        pb.shape.dim.add(size=d)
    assert C._entry(1, (3, 3, 3, 32), e['offset'], e['size'], e['crc32c']) == pb.SerializeToString(deterministic=True)
    raw = np.asarray(t['yolo2_darknet/conv0/weights']).tobytes()
    assert e['crc32c'] == T.masked_crc32c(raw) and e['size'] == len(raw)


def test_checkpoint_index_is_a_leveldb_table(tmp_path):
    """A second, minimal parser written from the LevelDB table format description: footer, index block, data blocks, restart
    arrays, block trailers; keys come out sorted and complete."""
    rng = np.random.RandomState(1)
    t = _tensors(rng)
    prefix = str(tmp_path / 'm')
    C.write(prefix, t, block_bytes=512)
    buf = open(prefix + '.index', 'rb').read()
    assert struct.unpack('<Q', buf[-8:])[0] == 0xdb4775248b80fb57 and len(buf) >= 48

    def varint(b, i):
        n = s = 0
        while True:
            c = b[i]; i += 1
            n |= (c & 127) << s; s += 7
            if c < 128:
                return n, i

    def block(off, size):
        blk = buf[off:off + size]
        assert buf[off + size] == 0                                              # uncompressed
        crc = T.crc32c(blk + b'\x00')
        assert struct.unpack('<I', buf[off + size + 1:off + size + 5])[0] == (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF
        nre = struct.unpack('<I', blk[-4:])[0]
        restarts = struct.unpack('<%dI' % nre, blk[-4 - 4 * nre:-4])
        end, i, key, out = len(blk) - 4 - 4 * nre, 0, b'', []
        while i < end:
            if i in restarts:
                start = i
            sh, i = varint(blk, i); ns, i = varint(blk, i); vl, i = varint(blk, i)
            if i - 3 <= start + 3 and sh != 0 and start == i:                    # (a restart entry shares nothing)
                raise AssertionError
            key = key[:sh] + blk[i:i + ns]; i += ns
            out.append((key, blk[i:i + vl])); i += vl
        return out, restarts

    f = buf[-48:]
    mo, i = varint(f, 0); ms, i = varint(f, i); io, i = varint(f, i); isz, i = varint(f, i)
    meta, _ = block(mo, ms)
    assert meta == []
    index, _ = block(io, isz)
    assert len(index) > 3                                                         # several data blocks
    keys = []
    for sep, handle in index:
        o, j = varint(handle, 0); n, j = varint(handle, j)
        ents, restarts = block(o, n)
        assert restarts[0] == 0 and ents[-1][0] <= sep
        keys += [k for k, _ in ents]
    assert keys == sorted(keys) and keys[0] == b'' and set(k.decode() for k in keys[1:]) == set(t)


def test_checkpoint_corruption_is_detected(tmp_path):
    rng = np.random.RandomState(2)
    prefix = str(tmp_path / 'm')
    C.write(prefix, _tensors(rng))
    data = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    data[100] ^= 1
    open(prefix + '.data-00000-of-00001', 'wb').write(data)
    with pytest.raises(IOError, match='corrupted'):
        C.read(prefix)
    C.write(prefix, _tensors(rng))
    idx = bytearray(open(prefix + '.index', 'rb').read())
    idx[10] ^= 1
    open(prefix + '.index', 'wb').write(idx)
    with pytest.raises(IOError):
        C.read(prefix)
    open(prefix + '.index', 'wb').write(b'not a table')
    with pytest.raises(IOError, match='magic'):
        C.read(prefix)


def test_latest_checkpoint_follows_the_state_file(tmp_path):
    d = str(tmp_path)
    for step in (5, 20, 100):
        C.write(os.path.join(d, 'model.ckpt-%d' % step), {'global_step': np.int64(step)})
    assert C.latest_checkpoint(d).endswith('model.ckpt-100')
    open(os.path.join(d, 'checkpoint'), 'w').write('model_checkpoint_path: "model.ckpt-20"\nall_model_checkpoint_paths: "model.ckpt-20"\n')
    assert C.latest_checkpoint(d).endswith('model.ckpt-20')


class _FakeEngine(object):
    """Just enough of engine.Engine for C.save / restore: variables by name, the flat parameter layout, set_variables."""

    def __init__(self):
        from yolo_tf_amd import graph as G
        self.graph = G.Graph()
        x = G.placeholder(self.graph, 'image', 32, 32)
        G.conv2d(x, 8, 3, scope='net/conv0')
        self.values = {v.name: np.full(v.shape, i + 1, np.float32) for i, v in enumerate(self.graph.variables.values())}
        self.param_offsets, off = {}, 0
        for v in self.graph.trainable():
            self.param_offsets[v.name] = (off, v.size)
            off += v.size
        self.n = off

    def get_variables(self):
        return dict(self.values)

    def set_variables(self, values, strict=True):
        self.values.update({k: np.asarray(v, np.float32) for k, v in values.items()})


class _FakeSession(object):
    def __init__(self):
        import torch
        import types
        self.engine = _FakeEngine()
        self.global_step = 0
        self.optimizer = types.SimpleNamespace(name='adam', slots=[torch.zeros(self.engine.n), torch.ones(self.engine.n)])


def test_saver_keeps_five_checkpoints_and_lists_them(tmp_path):
    """[TF-sem] tf.train.Saver(max_to_keep=5): old .index / .data files are pruned, the state file lists the kept ones."""
    s = _FakeSession()
    for step in (10, 20, 30, 40, 50, 60, 70):
        s.global_step = step
        C.save(str(tmp_path), s)
    kept = sorted(int(f.split('-')[1].split('.')[0]) for f in os.listdir(str(tmp_path)) if f.endswith('.index'))
    assert kept == [30, 40, 50, 60, 70]
    assert not [f for f in os.listdir(str(tmp_path)) if f.startswith('model.ckpt-10.') or f.startswith('model.ckpt-20.')]
    state = open(os.path.join(str(tmp_path), 'checkpoint')).read()
    assert state.startswith('model_checkpoint_path: "model.ckpt-70"')
    assert [int(l.split('-')[1].rstrip('"')) for l in state.splitlines() if l.startswith('all_model')] == kept
    assert C.checkpoint_step(C.latest_checkpoint(str(tmp_path))) == 70


def test_transfer_from_a_tf_checkpoint_takes_global_step_unless_excluded(tmp_path):
    """The reference's -t / -e transfer restores slim.get_variables_to_restore(exclude=...), which includes global_step: the learning
    rate schedule continues from the donor's step (both containers behave the same; train.py passes the session to both)."""
    donor = _FakeSession()
    donor.global_step = 1234
    prefix = C.save(str(tmp_path), donor)
    s = _FakeSession()
    s.optimizer.slots[0].fill_(7)
    assert C.restore(prefix, s, variables_only=True) == 1234
    assert s.global_step == 1234 and float(s.optimizer.slots[0][0]) == 7      # slots untouched by a transfer
    s2 = _FakeSession()
    C.restore(prefix, s2, exclude=['global_step'], variables_only=True)
    assert s2.global_step == 0
    s3 = _FakeSession()
    C.restore(prefix, s3)                                             # full resume: step and Adam slots
    assert s3.global_step == 1234 and float(s3.optimizer.slots[1][0]) == 1

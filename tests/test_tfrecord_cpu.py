"""the reference's TFRecord dataset cache (utils/data/cache.py:95-100, utils/data/__init__.py:28-47) read and
written without TensorFlow.  The framing is checked against published CRC32C vectors, the Example encoding against
google.protobuf's own serializer on the tf.train.Example schema (an independent implementation of the wire format)."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yolo_tf_amd.utils import tfrecord as T   # noqa: E402


def test_crc32c_known_answers():
    assert T.crc32c(b'') == 0
    assert T.crc32c(b'123456789') == 0xE3069283                       # the standard check value
    assert T.crc32c(bytes(32)) == 0x8A9136AA                          # RFC 3720 B.4
    assert T.crc32c(b'\xff' * 32) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    crc = T.crc32c(b'abc')
    assert T.masked_crc32c(b'abc') == (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _example_classes():
    """tf.train.Example's schema (tensorflow/core/example/{example,feature}.proto) built with protobuf's descriptor API."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    f = descriptor_pb2.FileDescriptorProto(name='example_schema.proto', package='tfx', syntax='proto3')
    FD = descriptor_pb2.FieldDescriptorProto

    def msg(name):
        m = f.message_type.add()
        m.name = name
        return m
    m = msg('BytesList'); m.field.add(name='value', number=1, label=FD.LABEL_REPEATED, type=FD.TYPE_BYTES)
    m = msg('FloatList'); m.field.add(name='value', number=1, label=FD.LABEL_REPEATED, type=FD.TYPE_FLOAT)
    m = msg('Int64List'); m.field.add(name='value', number=1, label=FD.LABEL_REPEATED, type=FD.TYPE_INT64)
    m = msg('Feature')
    m.oneof_decl.add(name='kind')
    m.field.add(name='bytes_list', number=1, label=FD.LABEL_OPTIONAL, type=FD.TYPE_MESSAGE, type_name='.tfx.BytesList', oneof_index=0)
    m.field.add(name='float_list', number=2, label=FD.LABEL_OPTIONAL, type=FD.TYPE_MESSAGE, type_name='.tfx.FloatList', oneof_index=0)
    m.field.add(name='int64_list', number=3, label=FD.LABEL_OPTIONAL, type=FD.TYPE_MESSAGE, type_name='.tfx.Int64List', oneof_index=0)
    m = msg('Features')
    e = m.nested_type.add(name='FeatureEntry')
    e.options.map_entry = True
    e.field.add(name='key', number=1, label=FD.LABEL_OPTIONAL, type=FD.TYPE_STRING)
    e.field.add(name='value', number=2, label=FD.LABEL_OPTIONAL, type=FD.TYPE_MESSAGE, type_name='.tfx.Feature')
    m.field.add(name='feature', number=1, label=FD.LABEL_REPEATED, type=FD.TYPE_MESSAGE, type_name='.tfx.Features.FeatureEntry')
    m = msg('Example'); m.field.add(name='features', number=1, label=FD.LABEL_OPTIONAL, type=FD.TYPE_MESSAGE, type_name='.tfx.Features')
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName('tfx.Example'))


def test_example_encoding_matches_protobuf():
    pytest.importorskip('google.protobuf')
    Example = _example_classes()
    cls = np.array([3, 17, 0, -2], np.int64)
    coord = np.random.RandomState(0).rand(4, 4).astype(np.float32) * 300
    mine = T.encode_sample('/data/VOC/JPEGImages/000123.jpg', (375, 500, 3), cls, coord)
    ex = Example()
    ex.features.feature['imagepath'].bytes_list.value.append(b'/data/VOC/JPEGImages/000123.jpg')
    ex.features.feature['imageshape'].int64_list.value.extend([375, 500, 3])
    ex.features.feature['objects'].bytes_list.value.extend([cls.tobytes(), coord.tobytes()])
    assert mine == ex.SerializeToString(deterministic=True)            # byte-identical to protobuf's serializer
    back = Example.FromString(mine)                                     # and protobuf parses ours
    assert list(back.features.feature['imageshape'].int64_list.value) == [375, 500, 3]
    path, shape, c2, b2 = T.decode_sample(ex.SerializeToString())
    assert path == '/data/VOC/JPEGImages/000123.jpg' and shape == (375, 500, 3)
    np.testing.assert_array_equal(c2, cls)
    np.testing.assert_array_equal(b2, coord)
    # float lists and negative int64 (ten-byte varints) round-trip too
    ex2 = Example()
    ex2.features.feature['f'].float_list.value.extend([1.5, -2.25])
    ex2.features.feature['i'].int64_list.value.extend([-1, 2 ** 40])
    d = T.decode_example(ex2.SerializeToString())
    assert d['f'] == ('float', [1.5, -2.25]) and d['i'] == ('int64', [-1, 2 ** 40])
    assert T.encode_example({'f': ('float', [1.5, -2.25]), 'i': ('int64', [-1, 2 ** 40])}) == ex2.SerializeToString(deterministic=True)


def test_cache_round_trip_with_jpegs(tmp_path):
    Image = pytest.importorskip('PIL.Image')
    rng = np.random.RandomState(1)
    samples = []
    for i, (h, w) in enumerate([(48, 64), (75, 50), (33, 33)]):
        arr = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        p = str(tmp_path / ('img%d.jpg' % i))
        Image.fromarray(arr).save(p, quality=95)
        k = i + 1
        x0, y0 = rng.uniform(0, w / 2, k), rng.uniform(0, h / 2, k)
        samples.append((p, (h, w, 3), rng.randint(0, 20, k), np.stack([x0, y0, x0 + 5, y0 + 7], 1)))
    cache = str(tmp_path / 'train.tfrecord')
    T.write_cache(cache, samples)
    recs = list(T.read_cache(cache))
    assert len(recs) == 3
    for (p, shape, c, b), (p0, s0, c0, b0) in zip(recs, samples):
        assert p == p0 and shape == s0
        np.testing.assert_array_equal(c, np.asarray(c0, np.int64))
        np.testing.assert_array_equal(b, np.asarray(b0, np.float32))
    images, objects = T.load_dataset(cache)
    assert [im.shape for im in images] == [(48, 64, 3), (75, 50, 3), (33, 33, 3)] and all(im.dtype == np.uint8 for im in images)
    assert objects[2][0].dtype == np.int32 and objects[2][1].shape == (3, 4)
    # framing: record = u64 length, masked crc of it, payload, masked crc of the payload; corruption is detected
    raw = open(cache, 'rb').read()
    (n0,) = struct.unpack('<Q', raw[:8])
    assert struct.unpack('<I', raw[8:12])[0] == T.masked_crc32c(raw[:8]) and n0 == len(T.encode_sample(*samples[0]))
    bad = bytearray(raw)
    bad[20] ^= 0x40
    open(cache, 'wb').write(bytes(bad))
    with pytest.raises(IOError):
        list(T.read_cache(cache))
    open(cache, 'wb').write(raw[:-3])
    with pytest.raises(IOError):
        list(T.read_cache(cache))

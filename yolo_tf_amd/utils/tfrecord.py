"""The reference's dataset cache format, without TensorFlow.

`cache.py` of the reference writes one TFRecord file per profile (utils/data/cache.py:66-103,106-155); each record is a
serialized `tf.train.Example` with three features (cache.py:95-99, parsed back in utils/data/__init__.py:28-47):

    imagepath   bytes_list [path of the JPEG]
    imageshape  int64_list [height, width, channels]
    objects     bytes_list [int64 class ids .tobytes(), float32 [K,4] (xmin, ymin, xmax, ymax) pixels .tobytes()]

This module reads and writes exactly that: the TFRecord framing (length, masked CRC32C of the length, payload, masked
CRC32C of the payload -- the published TFRecord format) and the protobuf wire encoding of `Example` (Features =
map<string, Feature>, Feature = oneof {BytesList = 1, FloatList = 2, Int64List = 3}), hand-rolled: the schema is three
messages deep and fixed.  `load_dataset` decodes the JPEGs with PIL (the reference: tf.image.decode_jpeg(channels=3)) and
returns what `utils.augment.DeviceInputPipeline` takes.  Host-side I/O: nothing here touches the GPU.
"""
import os
import struct

import numpy as np

# ---------------------------------------------------------------- CRC32C (Castagnoli), masked as TFRecord does
_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ _POLY if _c & 1 else _c >> 1
    _TABLE.append(_c)
_TABLE = np.array(_TABLE, np.uint32)


def crc32c(data):
    """CRC32C of a bytes-like object: through the C-ABI library's yolo2_crc32c when it is built (checkpoints are hundreds of MB),
    else the byte-at-a-time table below (records and events are small)."""
    data = bytes(data) if not isinstance(data, (bytes, bytearray, memoryview)) else data
    if len(data) >= 4096:
        try:
            from .. import _lib
            import ctypes
            buf = (ctypes.c_char * len(data)).from_buffer_copy(data) if not isinstance(data, bytes) else data
            return int(_lib.load().yolo2_crc32c(buf, len(data), 0))
        except Exception:
            pass
    crc = 0xFFFFFFFF
    table = _TABLE
    for b in bytes(data):
        crc = int(table[(crc ^ b) & 0xFF]) ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def masked_crc32c(data):
    crc = crc32c(data)
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------- TFRecord framing
def write_records(path, payloads):
    with open(path, 'wb') as f:
        for p in payloads:
            head = struct.pack('<Q', len(p))
            f.write(head)
            f.write(struct.pack('<I', masked_crc32c(head)))
            f.write(p)
            f.write(struct.pack('<I', masked_crc32c(p)))


def read_records(path, verify=True):
    with open(path, 'rb') as f:
        while True:
            head = f.read(8)
            if not head:
                return
            if len(head) != 8:
                raise IOError('%s: truncated record header' % path)
            (length,) = struct.unpack('<Q', head)
            tail = f.read(4)
            if len(tail) != 4:
                raise IOError('%s: truncated record header' % path)
            (hcrc,) = struct.unpack('<I', tail)
            if verify and hcrc != masked_crc32c(head):
                raise IOError('%s: corrupted record length' % path)
            data = f.read(length)
            if len(data) != length:
                raise IOError('%s: truncated record' % path)
            tail = f.read(4)
            if len(tail) != 4:
                raise IOError('%s: truncated record' % path)
            (dcrc,) = struct.unpack('<I', tail)
            if verify and dcrc != masked_crc32c(data):
                raise IOError('%s: corrupted record payload' % path)
            yield data


# ---------------------------------------------------------------- protobuf wire format (varint / length-delimited only)
def _varint(n):
    n &= (1 << 64) - 1                       # int64 two's complement, as protobuf encodes negative int64
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _field(number, payload):
    return _varint(number << 3 | 2) + _varint(len(payload)) + payload


def _fields(buf):
    """Yields (field number, wire type, value) of one message; value = bytes for length-delimited, int for varint."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _read_varint(buf, pos)
        number, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError('unsupported wire type %d' % wt)
        yield number, wt, v


def encode_example(features):
    """features: {name: ('bytes', [b, ...]) | ('int64', [int, ...]) | ('float', [float, ...])} -> serialized tf.train.Example.
    Entries are emitted in sorted key order (any order parses; sorted = what a deterministic serializer writes)."""
    entries = b''
    for name in sorted(features):
        kind, values = features[name]
        if kind == 'bytes':
            feat = _field(1, b''.join(_field(1, bytes(v)) for v in values))
        elif kind == 'float':
            feat = _field(2, _field(1, struct.pack('<%df' % len(values), *values)))            # packed
        elif kind == 'int64':
            feat = _field(3, _field(1, b''.join(_varint(int(v)) for v in values)))              # packed
        else:
            raise ValueError(kind)
        entry = _field(1, name.encode()) + _field(2, feat)                                      # map entry: key = 1, value = 2
        entries += _field(1, entry)                                                             # Features.feature
    return _field(1, entries)                                                                   # Example.features


def decode_example(buf):
    """Serialized tf.train.Example -> {name: ('bytes'|'int64'|'float', list)}; accepts packed and unpacked lists."""
    out = {}
    for n, wt, feats in _fields(bytes(buf)):
        if n != 1 or wt != 2:
            continue
        for n2, wt2, entry in _fields(feats):
            if n2 != 1 or wt2 != 2:
                continue
            key, value = None, None
            for n3, _, v in _fields(entry):
                if n3 == 1:
                    key = v.decode()
                elif n3 == 2:
                    value = v
            kind, vals = 'bytes', []
            for n4, _, lst in _fields(value or b''):
                if n4 == 1:
                    kind, vals = 'bytes', [v for k, _, v in _fields(lst) if k == 1]
                elif n4 == 2:
                    kind = 'float'
                    for k, wt5, v in _fields(lst):
                        if k == 1:
                            vals.extend(struct.unpack('<%df' % (len(v) // 4), v) if wt5 == 2 else struct.unpack('<f', v))
                elif n4 == 3:
                    kind = 'int64'
                    for k, wt5, v in _fields(lst):
                        if k != 1:
                            continue
                        if wt5 == 2:
                            pos = 0
                            while pos < len(v):
                                x, pos = _read_varint(v, pos)
                                vals.append(x - (1 << 64) if x >> 63 else x)
                        else:
                            vals.append(v - (1 << 64) if v >> 63 else v)
            out[key] = (kind, vals)
    return out


# ---------------------------------------------------------------- the reference's cache schema
def encode_sample(imagepath, imageshape, objects_class, objects_coord):
    """cache.py:95-99."""
    cls = np.asarray(objects_class, np.int64).reshape(-1)
    coord = np.asarray(objects_coord, np.float32).reshape(-1, 4)
    assert len(cls) == len(coord)
    return encode_example({
        'imagepath': ('bytes', [os.fsencode(imagepath)]),
        'imageshape': ('int64', [int(v) for v in imageshape]),
        'objects': ('bytes', [cls.tobytes(), coord.tobytes()]),
    })


def decode_sample(buf):
    """utils/data/__init__.py:31-44 (FixedLenFeature shapes enforced: imageshape [3], objects [2])."""
    ex = decode_example(buf)
    (path,) = ex['imagepath'][1]
    shape = ex['imageshape'][1]
    objs = ex['objects'][1]
    if len(shape) != 3 or len(objs) != 2:
        raise ValueError('record does not match the cache schema (imageshape [3], objects [2])')
    cls = np.frombuffer(objs[0], np.int64)
    coord = np.frombuffer(objs[1], np.float32).reshape(-1, 4)
    return os.fsdecode(path), tuple(int(v) for v in shape), cls, coord


def write_cache(path, samples):
    """samples: iterable of (imagepath, (h, w, c), classes, coords)."""
    write_records(path, (encode_sample(*s) for s in samples))


def read_cache(paths):
    if isinstance(paths, str):
        paths = [paths]
    for p in paths:
        for rec in read_records(p):
            yield decode_sample(rec)


def load_dataset(paths, limit=None):
    """Cache files -> (images: list of uint8 [h,w,3], objects: list of (classes, coords)) for DeviceInputPipeline.
    JPEG decode: PIL, converted to 3 channels like tf.image.decode_jpeg(channels=3)."""
    from PIL import Image
    images, objects = [], []
    for imagepath, shape, cls, coord in read_cache(paths):
        with Image.open(imagepath) as im:
            a = np.asarray(im.convert('RGB'), np.uint8)
        if a.shape[:2] != tuple(shape[:2]):
            raise ValueError('%s: decoded %s, cache says %s' % (imagepath, a.shape, shape))
        images.append(a)
        objects.append((cls.astype(np.int32), coord.copy()))
        if limit is not None and len(images) >= limit:
            break
    return images, objects

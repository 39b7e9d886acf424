"""TensorBoard event files without TensorFlow.

The reference hands ``slim.learning.train`` a ``tf.summary.FileWriter(os.path.join(logdir, args.logname))`` (train.py:141-145)
and summarises the scalars its ``[summary] scalar`` pattern selects (config.ini:63, train.py:31-41): ``total_loss`` and
``total_loss/objectives/{iou_best,iou_normal,coords,prob}``.  This module writes the same kind of file -- TFRecord framing
(utils/tfrecord.py) around serialized ``tensorflow.Event`` protos -- for exactly those scalars, hand-rolling the three tiny
messages it needs:

    Event   { double wall_time = 1; int64 step = 2; oneof { string file_version = 3; Summary summary = 5; } }
    Summary { repeated Value value = 1; }      Value { string tag = 1; float simple_value = 2; }

TensorBoard reads it like any ``events.out.tfevents.*`` file.  Image and histogram summaries of the reference (train.py:44-60)
are not produced.  Host-side I/O only."""
import os
import socket
import struct
import time

from . import tfrecord

SCALAR_TAGS = ('total_loss', 'total_loss/objectives/iou_best', 'total_loss/objectives/iou_normal', 'total_loss/objectives/coords',
               'total_loss/objectives/prob')


def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field, payload):          # length-delimited field
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def encode_event(wall_time, step=None, file_version=None, scalars=None):
    ev = _varint((1 << 3) | 1) + struct.pack('<d', float(wall_time))
    if step is not None and step != 0:
        ev += _varint((2 << 3) | 0) + _varint(int(step))
    if file_version is not None:
        ev += _ld(3, file_version.encode())
    if scalars is not None:
        summary = b''
        for tag, value in scalars:
            summary += _ld(1, _ld(1, tag.encode()) + _varint((2 << 3) | 5) + struct.pack('<f', float(value)))
        ev += _ld(5, summary)
    return ev


def decode_event(buf):
    """Inverse of :func:`encode_event` for the fields it writes (tests, and reading files back)."""
    def fields(b):
        i = 0
        while i < len(b):
            key, i = _read_varint(b, i)
            f, wt = key >> 3, key & 7
            if wt == 0:
                v, i = _read_varint(b, i)
            elif wt == 1:
                v, i = b[i:i + 8], i + 8
            elif wt == 5:
                v, i = b[i:i + 4], i + 4
            elif wt == 2:
                n, i = _read_varint(b, i)
                v, i = b[i:i + n], i + n
            else:
                raise ValueError('wire type %d' % wt)
            yield f, wt, v

    out = {'step': 0, 'scalars': []}
    for f, wt, v in fields(buf):
        if f == 1:
            out['wall_time'] = struct.unpack('<d', v)[0]
        elif f == 2:
            out['step'] = v
        elif f == 3:
            out['file_version'] = v.decode()
        elif f == 5:
            for f2, _, val in fields(v):
                if f2 == 1:
                    tag, simple = None, None
                    for f3, _, x in fields(val):
                        if f3 == 1:
                            tag = x.decode()
                        elif f3 == 2:
                            simple = struct.unpack('<f', x)[0]
                    out['scalars'].append((tag, simple))
    return out


def _read_varint(b, i):
    shift, n = 0, 0
    while True:
        c = b[i]
        i += 1
        n |= (c & 0x7F) << shift
        if not c & 0x80:
            return n, i
        shift += 7


class FileWriter(object):
    """``tf.summary.FileWriter(logdir)`` for scalar summaries: appends to ``<logdir>/events.out.tfevents.<time>.<host>``."""

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        now = time.time()
        self.path = os.path.join(logdir, 'events.out.tfevents.%010d.%s' % (int(now), socket.gethostname()))
        self._f = open(self.path, 'ab')
        self._write(encode_event(now, file_version='brain.Event:2'))

    def _write(self, payload):
        head = struct.pack('<Q', len(payload))
        self._f.write(head + struct.pack('<I', tfrecord.masked_crc32c(head)) + payload + struct.pack('<I', tfrecord.masked_crc32c(payload)))

    def add_scalars(self, step, values, wall_time=None):
        """values: {tag: float} or [(tag, float)]."""
        items = list(values.items()) if isinstance(values, dict) else list(values)
        self._write(encode_event(time.time() if wall_time is None else wall_time, step=step, scalars=items))

    def add_training_summary(self, step, fetched):
        """The five scalars of the reference's [summary] section from TrainSession.fetch()'s dict."""
        self.add_scalars(step, [('total_loss', fetched['total_loss'])] +
                         [('total_loss/objectives/' + k, fetched[k]) for k in ('iou_best', 'iou_normal', 'coords', 'prob')])

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def read_events(path):
    return [decode_event(p) for p in tfrecord.read_records(path)]

"""Darknet ``.weights`` import / export for the YOLOv2 graphs (counterpart of the reference's
parse_darknet_yolo2.py:58-117, which converts such a file into a TF checkpoint).

File format (as the reference reads it): a 16-byte header ``int32 major, minor, revision, seen`` (:79), then per
convolution in network order (conv0 ... convN, the final 1x1 last, :71-76) float32 blocks in the order
``biases | beta, gamma, moving_mean, moving_variance, weights`` (only the ones the layer has, :84), the weights stored
OIHW (:95) and converted to HWIO (:96).  The final layer's output channels are per anchor ``(x, y, w, h, obj, cls...)``
in Darknet and ``(obj, x, y, w, h, cls...)`` here, so its weights/biases are permuted (:34-48, :101).
Pure host-side file handling (NumPy); tensors go to the device through ``Engine.set_variables``.
"""
import struct

import numpy as np

_ORDER = ('biases', 'BatchNorm/beta', 'BatchNorm/gamma', 'BatchNorm/moving_mean', 'BatchNorm/moving_variance', 'weights')


def _anchor_perm(num_anchors, per_anchor, inverse=False):
    """Index permutation of the final layer's output channels: Darknet (x,y,w,h,obj,cls..) -> (obj,x,y,w,h,cls..)."""
    one = [4, 0, 1, 2, 3] + list(range(5, per_anchor))
    if inverse:
        one = list(np.argsort(one))
    return np.concatenate([np.asarray(one) + a * per_anchor for a in range(num_anchors)])


def transpose_weights(weights, num_anchors, inverse=False):
    return weights[..., _anchor_perm(num_anchors, weights.shape[-1] // num_anchors, inverse)]


def transpose_biases(biases, num_anchors, inverse=False):
    return biases[_anchor_perm(num_anchors, biases.shape[0] // num_anchors, inverse)]


def _layers(graph):
    """[(scope, {suffix: Variable})] in file order; the head (no BatchNorm) is last by construction of the plugins."""
    out = []
    for op in graph.ops:
        if op['kind'] != 'conv':
            continue
        scope = op['name']
        out.append((scope, {k[len(scope) + 1:]: v for k, v in graph.variables.items() if k.startswith(scope + '/')}))
    return out


def load(path, graph, num_anchors):
    """Returns (header, {variable name: float32 array}) for every variable of ``graph`` found in the file.
    Raises ValueError if the file is shorter than the graph needs; ``header['remaining']`` counts unread bytes
    (the reference only warns about them, :116-117)."""
    with open(path, 'rb') as f:
        raw = f.read()
    major, minor, revision, seen = struct.unpack_from('4i', raw, 0)
    pos = 16
    values = {}
    layers = _layers(graph)
    for li, (scope, var) in enumerate(layers):
        for suffix in _ORDER:
            if suffix not in var:
                continue
            v = var[suffix]
            n = v.size
            if pos + 4 * n > len(raw):
                raise ValueError('%s: file ends inside %s/%s' % (path, scope, suffix))
            p = np.frombuffer(raw, np.float32, n, pos).copy()
            pos += 4 * n
            if suffix == 'weights':
                k1, k2, cin, cout = v.shape
                p = p.reshape(cout, cin, k1, k2).transpose(2, 3, 1, 0)      # OIHW -> HWIO
            values[v.name] = np.ascontiguousarray(p.reshape(v.shape))
        if li == len(layers) - 1:                                           # detection head: per-anchor channel order
            values[var['weights'].name] = np.ascontiguousarray(transpose_weights(values[var['weights'].name], num_anchors))
            values[var['biases'].name] = np.ascontiguousarray(transpose_biases(values[var['biases'].name], num_anchors))
    return {'major': major, 'minor': minor, 'revision': revision, 'seen': seen, 'remaining': len(raw) - pos}, values


def save(path, graph, values, num_anchors, header=(0, 1, 0, 0)):
    """Inverse of :func:`load`."""
    layers = _layers(graph)
    with open(path, 'wb') as f:
        f.write(struct.pack('4i', *header))
        for li, (scope, var) in enumerate(layers):
            for suffix in _ORDER:
                if suffix not in var:
                    continue
                p = np.asarray(values[var[suffix].name], np.float32)
                if li == len(layers) - 1:
                    p = transpose_weights(p, num_anchors, inverse=True) if suffix == 'weights' else transpose_biases(p, num_anchors, inverse=True)
                if suffix == 'weights':
                    p = p.transpose(3, 2, 0, 1)                               # HWIO -> OIHW
                f.write(np.ascontiguousarray(p, np.float32).tobytes())

"""Symbolic NHWC graph built by the inference plugins.

The reference's plugins (model/yolo2/inference.py) build a TF-1 graph once with tf.contrib.slim
layer calls, then a session executes it.  Here a plugin is still ``fn(net, classes, num_anchors,
training) -> (scope, net)``: it calls the layer functions below on a symbolic ``Tensor`` and they
append ops to a ``Graph``; ``engine.Engine`` binds the graph to device buffers and drives the HIP
kernels (forward, backward, optimizer).  Variable names follow the reference's TF scopes
(``yolo2_darknet/conv3/BatchNorm/gamma`` ...; cf. parse_darknet_yolo2.py:71) so checkpoints keyed
by those names map 1:1.
"""
from collections import OrderedDict

import numpy as np


def pad8(c):
    return (c + 7) // 8 * 8


class Tensor(object):
    """Activation [B, H, W, C] with pixel stride ``ld`` (batch is bound by the engine)."""

    def __init__(self, graph, name, h, w, c, ld=None):
        self.graph = graph
        self.name = name
        self.h, self.w, self.c = h, w, c
        self.ld = ld if ld is not None else pad8(c)
        self.base = None          # (Tensor, channel offset) when this tensor lives inside a concat buffer
        self.flat_of = None       # Tensor whose [h, w, c] pixels this [1, 1, h*w*c] tensor re-reads (slim.layers.flatten)
        self.producer = None
        self.n_consumers = 0

    def get_shape(self):
        return (None, self.h, self.w, self.c)

    def storage(self):
        """Resolves aliases: returns (root tensor, channel offset, ld)."""
        if self.flat_of is not None:
            r, off, _ = self.flat_of.storage()
            assert off == 0
            return r, 0, self.ld
        t, off = self, 0
        while t.base is not None:
            off += t.base[1]
            t = t.base[0]
        return t, off, t.ld

    def __repr__(self):
        return 'Tensor(%s, %dx%dx%d)' % (self.name, self.h, self.w, self.c)


class Variable(object):
    def __init__(self, name, shape, init, trainable=True):
        self.name, self.shape, self.init, self.trainable = name, tuple(shape), init, trainable

    @property
    def size(self):
        return int(np.prod(self.shape))


class Graph(object):
    def __init__(self):
        self.ops = []
        self.variables = OrderedDict()
        self.tensors = []
        self.inputs = {}

    def tensor(self, name, h, w, c, ld=None):
        t = Tensor(self, name, h, w, c, ld)
        self.tensors.append(t)
        return t

    def variable(self, name, shape, init, trainable=True):
        if name in self.variables:
            raise ValueError('variable %s already exists' % name)
        v = Variable(name, shape, init, trainable)
        self.variables[name] = v
        return v

    def add(self, op):
        for k in op.get('inputs', ()):
            k.n_consumers += 1
        if op.get('out') is not None:
            op['out'].producer = op
        self.ops.append(op)
        return op

    def trainable(self):
        return [v for v in self.variables.values() if v.trainable]


def placeholder(graph, name, height, width, channels=3):
    """Image input; stored 8 channels wide (3 real) so every pixel is one 16-byte bf16 vector."""
    t = graph.tensor(name, height, width, channels, ld=8)
    graph.inputs[name] = t
    return t


# ---- initialisers ([TF-sem] slim defaults) -----------------------------------------------------

def xavier_uniform(rng, shape):
    k = shape[0] * shape[1]
    lim = np.sqrt(6.0 / (k * shape[2] + k * shape[3]))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def truncated_normal(stddev):
    def init(rng, shape):
        x = rng.randn(*shape)
        bad = np.abs(x) > 2
        while bad.any():                      # tf.truncated_normal re-draws beyond 2 sigma
            x[bad] = rng.randn(int(bad.sum()))
            bad = np.abs(x) > 2
        return (x * stddev).astype(np.float32)
    return init


def xavier_uniform_fc(rng, shape):
    """[TF-sem] slim.fully_connected default initialiser: Xavier uniform over (fan_in, fan_out)."""
    lim = np.sqrt(6.0 / (shape[0] + shape[1]))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def zeros(rng, shape):
    return np.zeros(shape, np.float32)


def ones(rng, shape):
    return np.ones(shape, np.float32)


# ---- layer functions (the slim-shaped API plugins are written against) -------------------------

def conv2d(net, num_outputs, kernel_size=3, scope=None, batch_norm=True, activation=True, center=True,
           weights_initializer=xavier_uniform):
    """slim.layers.conv2d + normalizer_fn=batch_norm + activation_fn=leaky_relu as ONE op
    (reference arg_scope model/yolo2/inference.py:69).  Stride 1, SAME, no bias when normalised;
    ``batch_norm=False, activation=False`` gives the final biased 1x1 (:118).  ``center=False`` is the
    `_darknet`/`_tiny` variant: no beta, explicit `biases` after the normalisation (:64-65)."""
    g = net.graph
    num_outputs = int(num_outputs)                    # the reference passes `channels / 2` (a float)
    k = int(kernel_size[0] if isinstance(kernel_size, (list, tuple)) else kernel_size)
    assert k in (1, 3)
    cin = net.c
    w = g.variable(scope + '/weights', (k, k, cin, num_outputs), weights_initializer)
    op = {'kind': 'conv', 'name': scope, 'inputs': [net], 'x': net, 'ksize': k, 'cin': cin, 'cout': num_outputs,
          'bn': bool(batch_norm), 'act': bool(activation), 'weights': w}
    if batch_norm:
        op['gamma'] = g.variable(scope + '/BatchNorm/gamma', (num_outputs,), ones)
        if center:
            op['beta'] = g.variable(scope + '/BatchNorm/beta', (num_outputs,), zeros)
        else:
            op['beta'] = g.variable(scope + '/biases', (num_outputs,), zeros)
        op['moving_mean'] = g.variable(scope + '/BatchNorm/moving_mean', (num_outputs,), zeros, trainable=False)
        op['moving_variance'] = g.variable(scope + '/BatchNorm/moving_variance', (num_outputs,), ones, trainable=False)
        op['y'] = g.tensor(scope + '/convolution', net.h, net.w, num_outputs)
        assert activation, 'normalised convolutions are always followed by leaky_relu on this path'
    else:
        # no normaliser: slim adds `biases` (zeros) and applies the activation, if any, after the bias add -- the YOLOv1 stack
        # (reference model/yolo/inference.py:27-50) and the biased final 1x1 of YOLOv2
        op['biases'] = g.variable(scope + '/biases', (num_outputs,), zeros)
    op['out'] = g.tensor(scope + ('/leaky_relu' if activation else '/BiasAdd'), net.h, net.w, num_outputs)
    g.add(op)
    return op['out']


def flatten(net, scope=None):
    """slim.layers.flatten on NHWC: [B, h, w, c] -> [B, h*w*c] in (h, w, c) order (reference model/yolo/inference.py:54) -- the same
    bytes, re-read as one pixel of h*w*c channels (needs an unpadded pixel stride)."""
    assert net.base is None and net.ld == net.c, 'flatten needs a dense tensor'
    g = net.graph
    out = g.tensor(scope or (net.name + '/flatten'), 1, 1, net.h * net.w * net.c, ld=net.h * net.w * net.c)
    out.flat_of = net
    g.add({'kind': 'flatten', 'name': out.name, 'inputs': [net], 'x': net, 'out': out})
    return out


def fully_connected(net, num_outputs, scope=None, activation=True, weights_regularizer=0.0, weights_initializer=xavier_uniform_fc):
    """slim.layers.fully_connected (reference model/yolo/inference.py:55-61): weights [in, out] + biases, optional leaky_relu,
    optional slim.l2_regularizer(scale) on the weights.  Executed as a 1x1 convolution over the one-pixel image."""
    assert net.h == 1 and net.w == 1
    g = net.graph
    num_outputs = int(num_outputs)
    w = g.variable(scope + '/weights', (net.c, num_outputs), weights_initializer)
    op = {'kind': 'conv', 'name': scope, 'inputs': [net], 'x': net, 'ksize': 1, 'cin': net.c, 'cout': num_outputs, 'bn': False,
          'act': bool(activation), 'weights': w, 'l2': float(weights_regularizer), 'fc': True}
    op['biases'] = g.variable(scope + '/biases', (num_outputs,), zeros)
    op['out'] = g.tensor(scope + ('/leaky_relu' if activation else '/BiasAdd'), 1, 1, num_outputs)
    g.add(op)
    return op['out']


def dropout(net, keep_prob=0.5, is_training=False, scope=None):
    """slim.layers.dropout (reference model/yolo/inference.py:57,60): identity at inference; in training x * mask / keep_prob with
    mask ~ Bernoulli(keep_prob) drawn on the device per step."""
    if not is_training:
        return net
    g = net.graph
    out = g.tensor(scope, net.h, net.w, net.c)
    g.add({'kind': 'dropout', 'name': scope, 'inputs': [net], 'x': net, 'out': out, 'keep_prob': float(keep_prob)})
    return out


def max_pool2d(net, stride=2, scope=None):
    """slim.layers.max_pool2d, kernel 2x2, padding SAME (model/yolo2/inference.py:38,42,74,83,96)."""
    g = net.graph
    if stride == 2:
        assert net.h % 2 == 0 and net.w % 2 == 0
        oh, ow = net.h // 2, net.w // 2
    else:
        assert stride == 1
        oh, ow = net.h, net.w
    out = g.tensor(scope, oh, ow, net.c)
    g.add({'kind': 'pool', 'name': scope, 'inputs': [net], 'x': net, 'out': out, 'stride': stride})
    return out


def identity(net, name):
    """tf.identity: naming only (passthrough tap, model/yolo2/inference.py:95)."""
    return net


def concat(values, axis=3, name='concat'):
    """tf.concat on channels (model/yolo2/inference.py:116): no data movement -- the operands are
    re-homed as channel slices of one buffer, so their producers write straight into it."""
    assert axis == 3
    g = values[0].graph
    h, w = values[0].h, values[0].w
    out = g.tensor(name, h, w, sum(v.c for v in values))
    off = 0
    for v in values:
        assert (v.h, v.w) == (h, w) and v.base is None and v.c % 8 == 0
        assert v.n_consumers == 0, 'concat operands must not have other consumers'
        v.base = (out, off)
        off += v.c
    g.add({'kind': 'concat', 'name': name, 'inputs': list(values), 'out': out})
    return out

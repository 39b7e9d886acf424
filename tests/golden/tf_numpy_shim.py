"""A NumPy-backed stand-in for the ~40 TensorFlow-1.0 / tf.contrib.slim entry points the reference's model code uses,
so that the reference's OWN source for the decode, the objectives, reorg and the network topology
(/root/reference/model/yolo2/__init__.py:28-94, model/yolo/__init__.py:37-100, model/yolo2/function.py:22-47,
model/yolo2/inference.py:25-120, model/yolo/inference.py:23-64) can be executed in the build container and its results
committed as fixtures (tests/golden/make_golden.py).  Generator-side tooling only: nothing in the product, the oracle or
the tests imports it, and it never travels without /root/reference being present.

What is pinned by running the reference under this shim, and what is not:
  * pinned: everything the reference's Python decides -- tensor slicing / reshapes / transposes, the order and operands of
    every arithmetic op of Model and Objectives, layer order, channel counts, kernel sizes, strides, scope (variable) names,
    which layers carry batch norm / an activation / an L2 regulariser, where the passthrough is tapped, the concat order;
  * NOT pinned ([TF-sem], restated here from the TF-1.0 documentation): the arithmetic INSIDE each elementary op
    (NumPy float32 kernels instead of Eigen's: sigmoid, exp, softmax, reductions), SAME-padding rules of conv / pool,
    slim.batch_norm's formula and default decay, variable naming inside a slim layer scope (weights, biases, BatchNorm/*).

Graph model: a Tensor holds the function that computes it and its inputs; it is evaluated eagerly at construction (so
static shapes exist, placeholders evaluate as zeros) and re-evaluated by Session.run(fetches, feed_dict).
"""
import contextlib
import sys
import types

import numpy as np

LOG = []            # one dict per slim layer call / structural op, in call order
VARIABLES = {}      # name -> np.ndarray (creation order preserved)
VAR_INFO = []       # [{'name', 'shape', 'kind'}] in creation order
UPDATES = {}        # moving-average updates of the last training-mode construction
COLLECTIONS = {}
_REGISTRY = {}      # 'scope/name:0' -> Tensor
_SCOPES = []
_VALUE_FN = [None]  # callable(name, shape, kind) -> array: how variables get their values


def reset(value_fn=None):
    del LOG[:], VAR_INFO[:], _SCOPES[:]
    VARIABLES.clear(), UPDATES.clear(), COLLECTIONS.clear(), _REGISTRY.clear()
    _VALUE_FN[0] = value_fn


# ------------------------------------------------------------------------------------------------ tensors
class _Shape(object):
    def __init__(self, dims):
        self.dims = list(dims)

    def as_list(self):
        return [int(d) for d in self.dims]

    def __getitem__(self, i):
        return self.dims[i]

    def __iter__(self):
        return iter(self.dims)


class Tensor(object):
    __array_priority__ = 1000.0
    __array_ufunc__ = None          # ndarray <op> Tensor defers to Tensor.__r<op>__

    def __init__(self, fn, inputs=(), name=None):
        self.fn, self.inputs = fn, tuple(inputs)
        self.value = np.asarray(fn(*[i.value for i in self.inputs]))
        self.name = _register(self, name) if name is not None else None

    def get_shape(self):
        return _Shape(self.value.shape)

    @property
    def dtype(self):
        return self.value.dtype

    def _eval(self, memo):
        if id(self) not in memo:
            memo[id(self)] = np.asarray(self.fn(*[i._eval(memo) for i in self.inputs]))
        return memo[id(self)]

    def __getitem__(self, idx):
        return Tensor(lambda a: a[idx], [self])

    def _bin(self, other, f, swap=False):
        o = other if isinstance(other, Tensor) else _const(other, self.value.dtype)     # [TF-sem] a Python / NumPy operand takes the tensor's dtype
        a, b = (o, self) if swap else (self, o)
        return Tensor(f, [a, b])

    def __add__(self, o): return self._bin(o, np.add)
    def __radd__(self, o): return self._bin(o, np.add, True)
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __rsub__(self, o): return self._bin(o, np.subtract, True)
    def __mul__(self, o): return self._bin(o, np.multiply)
    def __rmul__(self, o): return self._bin(o, np.multiply, True)
    def __truediv__(self, o): return self._bin(o, np.true_divide)
    def __rtruediv__(self, o): return self._bin(o, np.true_divide, True)
    def __neg__(self): return Tensor(np.negative, [self])


def _const(v, dtype=None):
    a = np.asarray(v, dtype=dtype)
    return Tensor(lambda: a)


def _t(x, like=None):
    return x if isinstance(x, Tensor) else _const(x, None if like is None else like.value.dtype)


def _full_name(name):
    if name.endswith('/'):
        return name[:-1]                       # the string a `with tf.name_scope(..) as name` block yields: absolute
    return '/'.join(_SCOPES + [name])


def _register(t, name):
    full = _full_name(name)
    _REGISTRY[full + ':0'] = t
    return full


@contextlib.contextmanager
def name_scope(name):
    _SCOPES.append(name)
    try:
        yield '/'.join(_SCOPES) + '/'
    finally:
        _SCOPES.pop()


def _unary(f):
    def op(x, name=None):
        return Tensor(f, [_t(x)], name)
    return op


def _sigmoid(a):
    return (a.dtype.type(1) / (a.dtype.type(1) + np.exp(-a))).astype(a.dtype)


def _softmax(a):
    e = np.exp(a - a.max(-1, keepdims=True))
    return (e / e.sum(-1, keepdims=True)).astype(a.dtype)


def _reduce(f):
    def op(x, axis=None, keep_dims=False, name=None):
        return Tensor(lambda a: f(a, axis=axis, keepdims=keep_dims).astype(a.dtype), [_t(x)], name)
    return op


def _binary(f):
    def op(x, y, name=None):
        x = _t(x, y if isinstance(y, Tensor) else None)
        y = _t(y, x)
        return Tensor(f, [x, y], name)
    return op


def _concat(values, axis, name=None):
    ts = [_t(v) for v in values]
    LOG.append({'op': 'concat', 'name': _full_name(name) if name else None, 'axis': int(axis), 'input_channels': [int(t.value.shape[-1]) for t in ts]})
    return Tensor(lambda *a: np.concatenate(a, axis=axis), ts, name)


class _Graph(object):
    def get_tensor_by_name(self, name):
        return _REGISTRY[name]


class _Placeholder(Tensor):
    def __init__(self, dtype, shape):
        self._fed = np.zeros(shape, dtype)
        Tensor.__init__(self, lambda: self._fed)

    def _eval(self, memo):
        return memo.get(id(self), self._fed)


class Session(object):
    def __enter__(self): return self
    def __exit__(self, *a): return False

    def run(self, fetches, feed_dict=None):
        memo = {}
        for ph, v in (feed_dict or {}).items():
            memo[id(ph)] = np.asarray(v, ph.value.dtype)
        single = isinstance(fetches, Tensor)
        out = [f._eval(memo) for f in ([fetches] if single else fetches)]
        return out[0] if single else out


# ------------------------------------------------------------------------------------------------ variables
def _variable(name, shape, kind):
    full = _full_name(name)
    shape = [int(s) for s in shape]
    if full not in VARIABLES:
        v = _VALUE_FN[0](full, shape, kind)
        assert list(v.shape) == shape, (full, v.shape, shape)
        VARIABLES[full] = np.asarray(v, np.float32)
        VAR_INFO.append({'name': full, 'shape': shape, 'kind': kind})
    a = VARIABLES[full]
    return Tensor(lambda: a)


# ------------------------------------------------------------------------------------------------ slim
_ARG_SCOPES = []


@contextlib.contextmanager
def arg_scope(fns, **kwargs):
    _ARG_SCOPES.append({getattr(f, '_key', f): kwargs for f in fns})
    try:
        yield
    finally:
        _ARG_SCOPES.pop()


def _scoped(f):
    def wrapper(*args, **kwargs):
        merged = {}
        for level in _ARG_SCOPES:
            merged.update(level.get(wrapper, {}))
        merged.update(kwargs)
        return f(*args, **merged)
    wrapper._key = wrapper
    wrapper.__name__ = f.__name__
    return wrapper


def _same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2, total - total // 2          # [TF-sem] SAME: the odd pixel of padding goes to the bottom / right


def _conv_same(x, w, stride):
    kh, kw, cin, cout = w.shape
    b, h, ww, _ = x.shape
    oh, pt, pb = _same_pad(h, kh, stride)
    ow, pl, pr = _same_pad(ww, kw, stride)
    xp = np.pad(x.astype(np.float64), ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    acc = np.zeros((b, oh, ow, cout), np.float64)
    w64 = w.astype(np.float64)
    for r in range(kh):
        for s in range(kw):
            patch = xp[:, r:r + (oh - 1) * stride + 1:stride, s:s + (ow - 1) * stride + 1:stride, :]
            acc += patch.reshape(-1, cin).dot(w64[r, s]).reshape(b, oh, ow, cout)
    return acc.astype(np.float32)


def _fn_name(f):
    return None if f is None else getattr(f, '__name__', str(f))


class _Regularizer(object):
    def __init__(self, kind, scale):
        self.kind, self.scale = kind, float(scale)


@_scoped
def conv2d(inputs, num_outputs, kernel_size=None, stride=1, padding='SAME', activation_fn='relu', normalizer_fn=None,
           weights_initializer=None, weights_regularizer=None, biases_initializer='zeros', scope=None, **unused):
    assert padding == 'SAME' and scope is not None
    x = _t(inputs)
    cin, cout = int(x.value.shape[-1]), int(num_outputs)
    assert cout == num_outputs                               # (the reference passes channels / 2: a float with an integral value)
    kh, kw = [int(k) for k in kernel_size]
    entry = {'op': 'conv2d', 'scope': scope, 'kernel_size': [kh, kw], 'stride': int(stride), 'padding': padding, 'in_channels': cin,
             'num_outputs': cout, 'activation_fn': _fn_name(activation_fn), 'normalizer_fn': _fn_name(normalizer_fn),
             'weights_initializer': _fn_name(weights_initializer), 'weights_regularizer': None if weights_regularizer is None else
             [weights_regularizer.kind, weights_regularizer.scale], 'in_shape': [int(d) for d in x.value.shape]}
    LOG.append(entry)
    with name_scope(scope):
        w = _variable('weights', [kh, kw, cin, cout], 'weights')
        net = Tensor(lambda a, f: _conv_same(a, f, int(stride)), [x, w])
        if normalizer_fn is not None:
            net = normalizer_fn(net)                         # [TF-sem] slim: a normalizer replaces the bias
        else:
            net = bias_add(net, _variable('biases', [cout], 'biases'))
        if activation_fn is not None:
            assert activation_fn != 'relu', 'every conv2d of the reference overrides the default activation'
            net = activation_fn(net)
    entry['out_shape'] = [int(d) for d in net.value.shape]
    return net


@_scoped
def batch_norm(inputs, decay=0.999, center=True, scale=False, epsilon=0.001, is_training=True, **unused):
    x = _t(inputs)
    c = int(x.value.shape[-1])
    LOG.append({'op': 'batch_norm', 'scope': '/'.join(_SCOPES), 'decay': float(decay), 'center': bool(center), 'scale': bool(scale),
                'epsilon': float(epsilon), 'is_training': bool(is_training)})
    with name_scope('BatchNorm'):                            # [TF-sem] slim's default scope and variable names
        beta = _variable('beta', [c], 'beta') if center else None
        gamma = _variable('gamma', [c], 'gamma') if scale else None
        mm = _variable('moving_mean', [c], 'moving_mean')
        mv = _variable('moving_variance', [c], 'moving_variance')
        prefix = '/'.join(_SCOPES)

    def f(a, *params):
        p = list(params)
        b = p.pop(0) if center else np.zeros(c, np.float32)
        g = p.pop(0) if scale else np.ones(c, np.float32)
        m_, v_ = p
        if is_training:
            a64 = a.astype(np.float64).reshape(-1, c)
            mean, var = a64.mean(0), a64.var(0)              # [TF-sem] tf.nn.moments: biased variance
            UPDATES[prefix + '/moving_mean'] = (m_ * decay + mean * (1 - decay)).astype(np.float32)
            UPDATES[prefix + '/moving_variance'] = (v_ * decay + var * (1 - decay)).astype(np.float32)
        else:
            mean, var = m_.astype(np.float64), v_.astype(np.float64)
        return ((a.astype(np.float64) - mean) / np.sqrt(var + epsilon) * g + b).astype(np.float32)
    return Tensor(f, [x] + [t for t in (beta, gamma) if t is not None] + [mm, mv])


@_scoped
def max_pool2d(inputs, kernel_size=None, stride=2, padding='VALID', scope=None, **unused):
    x = _t(inputs)
    kh, kw = [int(k) for k in kernel_size]
    s = int(stride)
    LOG.append({'op': 'max_pool2d', 'scope': scope, 'kernel_size': [kh, kw], 'stride': s, 'padding': padding, 'in_shape': [int(d) for d in x.value.shape]})
    assert padding == 'SAME'

    def f(a):
        b, h, w, c = a.shape
        oh, pt, pb = _same_pad(h, kh, s)
        ow, pl, pr = _same_pad(w, kw, s)
        ap = np.pad(a, ((0, 0), (pt, pb), (pl, pr), (0, 0)), constant_values=-np.inf)
        out = np.full((b, oh, ow, c), -np.inf, a.dtype)
        for r in range(kh):
            for q in range(kw):
                out = np.maximum(out, ap[:, r:r + (oh - 1) * s + 1:s, q:q + (ow - 1) * s + 1:s, :])
        return out
    return Tensor(f, [x])


@_scoped
def flatten(inputs, scope=None, **unused):
    x = _t(inputs)
    LOG.append({'op': 'flatten', 'scope': scope, 'in_shape': [int(d) for d in x.value.shape]})
    return Tensor(lambda a: a.reshape(a.shape[0], -1), [x])


@_scoped
def fully_connected(inputs, num_outputs, activation_fn='relu', normalizer_fn=None, weights_regularizer=None, scope=None, **unused):
    x = _t(inputs)
    cin, cout = int(x.value.shape[-1]), int(num_outputs)
    LOG.append({'op': 'fully_connected', 'scope': scope, 'in_features': cin, 'num_outputs': cout, 'activation_fn': _fn_name(activation_fn),
                'normalizer_fn': _fn_name(normalizer_fn),
                'weights_regularizer': None if weights_regularizer is None else [weights_regularizer.kind, weights_regularizer.scale]})
    assert normalizer_fn is None
    with name_scope(scope):
        w = _variable('weights', [cin, cout], 'weights')
        b = _variable('biases', [cout], 'biases')
        net = Tensor(lambda a, f, c: (a.astype(np.float64).dot(f.astype(np.float64)) + c).astype(np.float32), [x, w, b])
        if activation_fn is not None:
            assert activation_fn != 'relu'
            net = activation_fn(net)
    return net


@_scoped
def dropout(inputs, keep_prob=0.5, is_training=True, scope=None, **unused):
    LOG.append({'op': 'dropout', 'scope': scope, 'keep_prob': float(keep_prob), 'is_training': bool(is_training)})
    assert not is_training, 'the fixtures run the inference graph (dropout = identity)'
    return _t(inputs)


def variable(name, shape=None, initializer=None, **unused):
    return _variable(name, [int(np.asarray(s.value if isinstance(s, Tensor) else s)) for s in shape], 'biases')


def bias_add(value, bias, name=None):
    return Tensor(lambda a, b: a + b, [_t(value), _t(bias)], name)


def l2_regularizer(scale):
    return _Regularizer('l2', scale)


def _initializer(tag):
    def make(*a, **k):
        def init():
            pass
        init.__name__ = tag if not k else '%s(%s)' % (tag, ', '.join('%s=%r' % kv for kv in sorted(k.items())))
        return init
    return make


# ------------------------------------------------------------------------------------------------ module objects
def install():
    """Puts the stand-in modules into sys.modules (tensorflow, tensorflow.contrib.slim, ...)."""
    tf = types.ModuleType('tensorflow')
    tf.reshape = lambda t, shape, name=None: Tensor(lambda a: a.reshape([int(s) for s in shape]), [_t(t)], name)
    tf.transpose = lambda t, perm, name=None: Tensor(lambda a: a.transpose(perm), [_t(t)], name)
    tf.identity = _unary(lambda a: a)
    tf.exp = _unary(np.exp)
    tf.sqrt = _unary(np.sqrt)
    tf.square = _unary(np.square)
    tf.abs = _unary(np.abs)
    tf.to_float = _unary(lambda a: a.astype(np.float32))
    tf.maximum = _binary(np.maximum)
    tf.minimum = _binary(np.minimum)
    tf.truediv = _binary(np.true_divide)
    tf.multiply = _binary(np.multiply)
    tf.equal = _binary(np.equal)
    tf.reduce_prod = _reduce(np.prod)
    tf.reduce_max = _reduce(np.max)
    tf.reduce_sum = _reduce(np.sum)
    tf.expand_dims = lambda t, axis, name=None: Tensor(lambda a: np.expand_dims(a, axis), [_t(t)], name)
    tf.concat = _concat
    tf.name_scope = name_scope
    tf.shape = lambda t: [_const(d) for d in _t(t).value.shape]
    tf.constant = lambda v, dtype=None: _const(v, dtype)
    tf.placeholder = lambda dtype, shape=None: _Placeholder(dtype, shape)
    tf.Session = Session
    tf.uint8, tf.float32 = np.uint8, np.float32
    tf.get_default_graph = lambda: _Graph()
    tf.add_to_collection = lambda key, value: COLLECTIONS.setdefault(key, []).append(value)
    tf.GraphKeys = types.SimpleNamespace(LOSSES='losses')
    tf.zeros_initializer = _initializer('zeros_initializer')
    tf.truncated_normal_initializer = _initializer('truncated_normal_initializer')
    tf.nn = types.SimpleNamespace(sigmoid=_unary(_sigmoid), softmax=_unary(_softmax), bias_add=bias_add)
    slim = types.ModuleType('tensorflow.contrib.slim')
    slim.arg_scope = arg_scope
    slim.batch_norm = batch_norm
    slim.variable = variable
    slim.l2_regularizer = l2_regularizer
    slim.layers = types.SimpleNamespace(conv2d=conv2d, max_pool2d=max_pool2d, flatten=flatten, fully_connected=fully_connected, dropout=dropout)
    contrib = types.ModuleType('tensorflow.contrib')
    contrib.slim = slim
    tf.contrib = contrib
    python = types.ModuleType('tensorflow.python')
    client = types.ModuleType('tensorflow.python.client')
    device_lib = types.ModuleType('tensorflow.python.client.device_lib')
    python.client, client.device_lib = client, device_lib
    tf.python = python
    for name, mod in (('tensorflow', tf), ('tensorflow.contrib', contrib), ('tensorflow.contrib.slim', slim), ('tensorflow.python', python),
                      ('tensorflow.python.client', client), ('tensorflow.python.client.device_lib', device_lib)):
        sys.modules[name] = mod
    return tf, slim

"""CPU ORACLE -- test infrastructure only, never shipped, never the thing measured.

NumPy restatement of the YOLOv2 train+detect hot path of ruiminshen/yolo-tf.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product package ``yolo_tf_amd`` never does.

Pinning status (see DESIGN.md section "Oracle"):
  * PINNED against the reference itself (functions importable here without TensorFlow, golden
    vectors in tests/golden/ produced by tests/golden/make_golden.py from /root/reference):
    ``iou``, ``non_max_suppress``, ``transform_labels``, ``calc_cell_xy``,
    ``per_image_standardization``, ``reorg`` (the reference's own 4x4 known-answer test,
    model/yolo2/function.py:32-47).
  * PARITY UNPINNED by the reference (the arithmetic lives in TensorFlow 1.0 / tf.contrib.slim,
    unpinned version, not vendored under /root/reference, cannot run here): conv2d, batch_norm,
    max_pool2d, Model decode, Objectives, backward, optimizers.  These follow the published
    TF-1.0 semantics ([TF-sem] notes) and are cross-checked in tests against torch-CPU fp64
    autograd and analytic known answers, not against TensorFlow.

All tensors are NHWC; conv weights are HWIO ``[kh, kw, Cin, Cout]`` (parse_darknet_yolo2.py:95-97).
Every function works in the dtype of its inputs (float32 for parity runs, float64 for
gradient cross-checks).
"""
import numpy as np

# --------------------------------------------------------------------------------------
# bf16 storage emulation (for checking the bf16 mode of the product: f32 arithmetic, values rounded to
# bfloat16 -- round-to-nearest-even on the upper 16 bits -- wherever the product stores a bf16 tensor)
# --------------------------------------------------------------------------------------

def bf16_round(x):
    a = np.ascontiguousarray(x, np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32).reshape(a.shape)


def _ident(x):
    return x


# --------------------------------------------------------------------------------------
# elementwise / layout ops
# --------------------------------------------------------------------------------------

def leaky_relu(x, alpha=.1):
    """model/yolo/function.py:21-24 -- max(x, alpha*x)."""
    return np.maximum(x, x.dtype.type(alpha) * x)


def leaky_relu_grad(x, dy, alpha=.1):
    """[TF-sem] tf.maximum(x, a*x) gradient: MaximumGrad routes to the first argument where
    x >= a*x, i.e. x >= 0 (ties at 0 take slope 1), else slope alpha."""
    return np.where(x >= 0, dy, x.dtype.type(alpha) * dy)


def reorg(x, stride=2):
    """model/yolo2/function.py:22-29 -- reshape/transpose/reshape; closed form
    out[b,y,x,(sy*2+sx)*C+c] = in[b,2y+sy,2x+sx,c]."""
    b, h, w, c = x.shape
    _h, _w = h // stride, w // stride
    t = x.reshape(b, _h, stride, _w, stride, c)
    t = t.transpose(0, 1, 3, 2, 4, 5)
    return t.reshape(b, _h, _w, stride * stride * c)


def reorg_grad(dy, stride=2):
    """Inverse permutation of :func:`reorg` (it is a pure move)."""
    b, _h, _w, cc = dy.shape
    c = cc // (stride * stride)
    t = dy.reshape(b, _h, _w, stride, stride, c)
    t = t.transpose(0, 1, 3, 2, 4, 5)
    return t.reshape(b, _h * stride, _w * stride, c)


# --------------------------------------------------------------------------------------
# convolution (slim.layers.conv2d call sites model/yolo2/inference.py:37-48,73-118)
# [TF-sem] stride 1, padding SAME, no dilation, cross-correlation (no kernel flip).
# --------------------------------------------------------------------------------------

def _shifted(x, dh, dw):
    """x shifted so that out[b,h,w] = x[b,h+dh,w+dw] with zero fill (SAME padding)."""
    b, h, w, c = x.shape
    out = np.zeros_like(x)
    hs, he = max(0, -dh), min(h, h - dh)
    ws, we = max(0, -dw), min(w, w - dw)
    if hs < he and ws < we:
        out[:, hs:he, ws:we, :] = x[:, hs + dh:he + dh, ws + dw:we + dw, :]
    return out


def conv2d(x, w):
    """y[b,h,w,k] = sum_{r,s,c} x[b,h+r-ph,w+s-pw,c] * W[r,s,c,k]; odd kernels, zero pad."""
    kh, kw, cin, cout = w.shape
    b, h, wd, _ = x.shape
    ph, pw = (kh - 1) // 2, (kw - 1) // 2
    y = np.zeros((b, h, wd, cout), dtype=x.dtype)
    for r in range(kh):
        for s in range(kw):
            xs = _shifted(x, r - ph, s - pw).reshape(-1, cin)
            y += (xs @ w[r, s]).reshape(b, h, wd, cout)
    return y


def conv2d_dgrad(dy, w):
    """dx[b,h,w,c] = sum_{r,s,k} dy[b,h-(r-ph),w-(s-pw),k] * W[r,s,c,k]."""
    kh, kw, cin, cout = w.shape
    b, h, wd, _ = dy.shape
    ph, pw = (kh - 1) // 2, (kw - 1) // 2
    dx = np.zeros((b, h, wd, cin), dtype=dy.dtype)
    for r in range(kh):
        for s in range(kw):
            ds = _shifted(dy, -(r - ph), -(s - pw)).reshape(-1, cout)
            dx += (ds @ w[r, s].T).reshape(b, h, wd, cin)
    return dx


def conv2d_wgrad(x, dy, kh, kw):
    """dW[r,s,c,k] = sum_{b,h,w} x[b,h+r-ph,w+s-pw,c] * dy[b,h,w,k]."""
    cin, cout = x.shape[-1], dy.shape[-1]
    ph, pw = (kh - 1) // 2, (kw - 1) // 2
    dw = np.zeros((kh, kw, cin, cout), dtype=x.dtype)
    d2 = dy.reshape(-1, cout)
    for r in range(kh):
        for s in range(kw):
            xs = _shifted(x, r - ph, s - pw).reshape(-1, cin)
            dw[r, s] = xs.T @ d2
    return dw


# --------------------------------------------------------------------------------------
# batch norm (closure model/yolo2/inference.py:62-66 -> slim.batch_norm(center, scale=True,
# epsilon=1e-5, is_training)).  [TF-sem] decay 0.999, tf.nn.moments (biased variance),
# y = gamma*(x-mean)/sqrt(var+eps)+beta, EMA with the biased batch variance.
# --------------------------------------------------------------------------------------
BN_EPS = 1e-5
BN_DECAY = 0.999


def bn_moments(x):
    n = x.shape[0] * x.shape[1] * x.shape[2]
    x2 = x.reshape(n, -1)
    acc = x2.astype(np.float64)
    mean = acc.mean(0)
    var = ((acc - mean) ** 2).mean(0)
    return mean.astype(x.dtype), var.astype(x.dtype)


def bn_apply(x, mean, var, gamma, beta, eps=BN_EPS):
    t = x.dtype.type
    inv = t(1) / np.sqrt(var + t(eps))
    return (x - mean) * (inv * gamma) + beta


def bn_ema(moving, batch, decay=BN_DECAY):
    """[TF-sem] assign_moving_average: moving -= (1-decay)*(moving-batch)."""
    t = moving.dtype.type
    return moving - (moving - batch) * t(1 - decay)


def bn_train_bwd(x, mean, var, gamma, dz, eps=BN_EPS):
    """Backward of y = gamma*xhat+beta with batch statistics; returns dx, dgamma, dbeta."""
    t = x.dtype.type
    n = x.shape[0] * x.shape[1] * x.shape[2]
    inv = t(1) / np.sqrt(var + t(eps))
    xhat = (x - mean) * inv
    dz64 = dz.reshape(n, -1).astype(np.float64)
    dbeta = dz64.sum(0).astype(x.dtype)
    dgamma = (dz64 * xhat.reshape(n, -1)).sum(0).astype(x.dtype)
    dx = (gamma * inv) * (dz - dbeta / t(n) - xhat * (dgamma / t(n)))
    return dx, dgamma, dbeta


# --------------------------------------------------------------------------------------
# max pool 2x2 (slim.layers.max_pool2d model/yolo2/inference.py:38,42,74,83,96; arg_scope
# padding='SAME').  [TF-sem] stride 2 on even extents needs no padding; the tiny model's
# stride-1 pool pads 0 before / 1 after and the pad never wins the max.  Backward routes the
# gradient to the first maximum in window scan order (row-major).
# --------------------------------------------------------------------------------------

def _pool_windows(x, stride):
    b, h, w, c = x.shape
    if stride == 2:
        assert h % 2 == 0 and w % 2 == 0
        xp = x
        oh, ow = h // 2, w // 2
    else:
        xp = np.full((b, h + 1, w + 1, c), -np.inf, dtype=x.dtype)
        xp[:, :h, :w, :] = x
        oh, ow = h, w
    win = np.stack([xp[:, dy:dy + oh * stride:stride, dx:dx + ow * stride:stride, :]
                    for dy in range(2) for dx in range(2)], axis=0)  # [4, b, oh, ow, c]
    return win, oh, ow


def max_pool(x, stride=2):
    win, _, _ = _pool_windows(x, stride)
    return win.max(0)


def max_pool_grad(x, dy, stride=2):
    b, h, w, c = x.shape
    win, oh, ow = _pool_windows(x, stride)
    arg = win.argmax(0)  # first max in scan order (dy-major, dx-minor)
    dx_full = np.zeros((b, h + 1, w + 1, c), dtype=dy.dtype)
    for k in range(4):
        ky, kx = k // 2, k % 2
        sel = np.where(arg == k, dy, dy.dtype.type(0))
        dx_full[:, ky:ky + oh * stride:stride, kx:kx + ow * stride:stride, :] += sel
    return dx_full[:, :h, :w, :]


# --------------------------------------------------------------------------------------
# network topology (restated from model/yolo2/inference.py; never imported from the product)
# --------------------------------------------------------------------------------------

def darknet_spec(classes, num_anchors):
    """model/yolo2/inference.py:61-120.  Returns a list of ops:
    ('conv', name, ksize, cout, bn) | ('pool', stride) | ('mark',) | ('reorg_concat',)."""
    ops = []
    idx = 0
    ch = 32
    for _ in range(2):  # :72-76
        ops += [('conv', 'conv%d' % idx, 3, ch, True), ('pool', 2)]
        idx += 1
        ch *= 2
    for _ in range(2):  # :77-85
        ops += [('conv', 'conv%d' % idx, 3, ch, True)]
        idx += 1
        ops += [('conv', 'conv%d' % idx, 1, ch // 2, True)]
        idx += 1
        ops += [('conv', 'conv%d' % idx, 3, ch, True), ('pool', 2)]
        idx += 1
        ch *= 2
    for k, c in ((3, ch), (1, ch // 2), (3, ch), (1, ch // 2), (3, ch)):  # :86-94
        ops += [('conv', 'conv%d' % idx, k, c, True)]
        idx += 1
    ops += [('mark',), ('pool', 2)]  # passthrough :95, pool :96
    ch *= 2
    for k, c in ((3, ch), (1, ch // 2), (3, ch), (1, ch // 2), (3, ch), (3, ch), (3, ch)):  # :100-112
        ops += [('conv', 'conv%d' % idx, k, c, True)]
        idx += 1
    ops += [('reorg_concat',)]  # :114-116, reorg output first
    ops += [('conv', 'conv%d' % idx, 3, ch, True)]  # :117
    ops += [('conv', 'conv', 1, num_anchors * (5 + classes), False)]  # :118
    return ops


def tiny_spec(classes, num_anchors):
    """model/yolo2/inference.py:25-50."""
    ops = []
    idx = 0
    ch = 16
    for _ in range(5):  # :36-40
        ops += [('conv', 'conv%d' % idx, 3, ch, True), ('pool', 2)]
        idx += 1
        ch *= 2
    ops += [('conv', 'conv%d' % idx, 3, ch, True), ('pool', 1)]  # :41-42
    idx += 1
    ch *= 2
    ops += [('conv', 'conv%d' % idx, 3, ch, True)]  # :45
    idx += 1
    ops += [('conv', 'conv%d' % idx, 3, ch, True)]  # :47
    ops += [('conv', 'conv', 1, num_anchors * (5 + classes), False)]  # :48
    return ops


SPECS = {'darknet': darknet_spec, 'tiny': tiny_spec}


def init_params(spec, cin=3, seed=0, dtype=np.float32, tiny=False):
    """[TF-sem] slim conv2d default init: Xavier-uniform weights (tiny: truncated normal 0.1,
    model/yolo2/inference.py:33), gamma=1 beta=0 moving_mean=0 moving_variance=1, biases=0."""
    rng = np.random.RandomState(seed)
    params = {}
    c = cin
    mark_c = None
    for op in spec:
        if op[0] == 'conv':
            _, name, k, cout, bn = op
            if tiny and bn:
                w = np.clip(rng.randn(k, k, c, cout), -2, 2) * 0.1
            else:
                lim = np.sqrt(6.0 / (k * k * c + k * k * cout))
                w = rng.uniform(-lim, lim, size=(k, k, c, cout))
            params[name + '/weights'] = w.astype(dtype)
            if bn:
                params[name + '/BatchNorm/gamma'] = np.ones(cout, dtype)
                params[name + '/BatchNorm/beta'] = np.zeros(cout, dtype)
                params[name + '/BatchNorm/moving_mean'] = np.zeros(cout, dtype)
                params[name + '/BatchNorm/moving_variance'] = np.ones(cout, dtype)
            else:
                params[name + '/biases'] = np.zeros(cout, dtype)
            c = cout
        elif op[0] == 'mark':
            mark_c = c
        elif op[0] == 'reorg_concat':
            c = mark_c * 4 + c
    return params


def trainable_names(params):
    return [k for k in params if not k.endswith(('moving_mean', 'moving_variance'))]


def network_forward(spec, params, x, training, quant=None):
    """Runs the op list; returns (net, caches).  Training uses batch statistics and returns
    the moving-average updates in caches['ema'] ([TF-sem] UPDATE_OPS run by create_train_op).
    ``quant`` (e.g. :func:`bf16_round`) is applied wherever the product's bf16 mode stores a tensor:
    input, filters, raw conv outputs, activations."""
    q = quant or _ident
    caches = []
    ema = {}
    net = q(x)
    mark = None
    for op in spec:
        if op[0] == 'conv':
            _, name, k, cout, bn = op
            w = q(params[name + '/weights'])
            y = conv2d(net, w)
            if bn:
                y = q(y)
                g = params[name + '/BatchNorm/gamma']
                bt = params[name + '/BatchNorm/beta']
                if training:
                    mean, var = bn_moments(y)
                    ema[name + '/BatchNorm/moving_mean'] = bn_ema(params[name + '/BatchNorm/moving_mean'], mean)
                    ema[name + '/BatchNorm/moving_variance'] = bn_ema(params[name + '/BatchNorm/moving_variance'], var)
                else:
                    mean = params[name + '/BatchNorm/moving_mean']
                    var = params[name + '/BatchNorm/moving_variance']
                z = bn_apply(y, mean, var, g, bt)
                out = q(leaky_relu(z))
                caches.append(('conv', name, net, y, mean, var, z))
            else:
                out = q(y + params[name + '/biases'])
                caches.append(('conv', name, net, None, None, None, None))
            net = out
        elif op[0] == 'pool':
            caches.append(('pool', op[1], net))
            net = max_pool(net, op[1])
        elif op[0] == 'mark':
            mark = net
            caches.append(('mark',))
        elif op[0] == 'reorg_concat':
            r = reorg(mark)
            caches.append(('reorg_concat', r.shape[-1]))
            net = np.concatenate([r, net], axis=3)
    return net, {'ops': caches, 'ema': ema}


def network_backward(spec, params, caches, dnet, quant=None):
    """Reverse sweep; returns gradients for every trainable variable (``quant``: see network_forward;
    applied to every activation gradient the product stores)."""
    q = quant or _ident
    grads = {}
    dmark = None
    dnet = q(dnet)
    for op, cache in zip(reversed(spec), reversed(caches['ops'])):
        if op[0] == 'conv':
            _, name, k, cout, bn = op
            _, _, xin, y, mean, var, z = cache
            w = q(params[name + '/weights'])
            if bn:
                dz = leaky_relu_grad(z, dnet)
                dy, dg, db = bn_train_bwd(y, mean, var, params[name + '/BatchNorm/gamma'], dz)
                dy = q(dy)
                grads[name + '/BatchNorm/gamma'] = dg
                grads[name + '/BatchNorm/beta'] = db
            else:
                dy = dnet
                grads[name + '/biases'] = dy.reshape(-1, cout).astype(np.float64).sum(0).astype(dy.dtype)
            grads[name + '/weights'] = conv2d_wgrad(xin, dy, k, k)
            dnet = q(conv2d_dgrad(dy, w))
        elif op[0] == 'pool':
            dnet = max_pool_grad(cache[2], dnet, cache[1])
        elif op[0] == 'mark':
            dnet = q(dnet + dmark)
        elif op[0] == 'reorg_concat':
            cr = cache[1]
            dmark = reorg_grad(dnet[..., :cr])
            dnet = dnet[..., cr:]
    return grads


# --------------------------------------------------------------------------------------
# decode head and loss (model/yolo2/__init__.py:28-94, model/yolo/__init__.py:29-34)
# --------------------------------------------------------------------------------------

def calc_cell_xy(cell_height, cell_width, dtype=np.float32):
    """model/yolo/__init__.py:29-34 -- cell_base[y,x,:] = [x,y]."""
    ys, xs = np.meshgrid(np.arange(cell_height), np.arange(cell_width), indexing='ij')
    return np.stack([xs, ys], -1).astype(dtype)


def sigmoid(x):
    return x.dtype.type(1) / (x.dtype.type(1) + np.exp(-x))


def softmax(x):
    e = np.exp(x - x.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True)


def model_decode(net, classes, anchors, training=False):
    """model/yolo2/__init__.py:28-59.  net [B,ch,cw,A*(5+C)], anchors [A,2] (w,h) in cells."""
    t = net.dtype.type
    b, ch, cw, _ = net.shape
    cells = ch * cw
    a = len(anchors)
    anchors = np.asarray(anchors, net.dtype)
    inputs = net.reshape(b, cells, a, 5 + classes)                      # :32
    sig = sigmoid(inputs[..., :3])                                       # :36
    m = {'cell_width': cw, 'cell_height': ch, 'inputs': inputs}
    m['iou'] = sig[..., 0]                                               # :37
    m['offset_xy'] = sig[..., 1:3]                                       # :38
    m['wh'] = np.exp(inputs[..., 3:5]) * anchors.reshape(1, 1, a, 2)     # :40
    m['prob'] = softmax(inputs[..., 5:])                                 # :42
    m['areas'] = m['wh'][..., 0] * m['wh'][..., 1]                       # :43
    half = m['wh'] / t(2)                                                # :44
    m['offset_xy_min'] = m['offset_xy'] - half                           # :45
    m['offset_xy_max'] = m['offset_xy'] + half                           # :46
    m['wh01'] = m['wh'] / np.array([cw, ch], net.dtype).reshape(1, 1, 1, 2)  # :47
    m['wh01_sqrt'] = np.sqrt(m['wh01'])                                  # :48
    m['coords'] = np.concatenate([m['offset_xy'], m['wh01_sqrt']], -1)   # :49
    if not training:                                                     # :50-56
        cell_xy = calc_cell_xy(ch, cw, net.dtype).reshape(1, cells, 1, 2)
        m['xy'] = cell_xy + m['offset_xy']
        m['xy_min'] = cell_xy + m['offset_xy_min']
        m['xy_max'] = cell_xy + m['offset_xy_max']
        m['conf'] = m['iou'][..., None] * m['prob']
    return m


OBJECTIVE_KEYS = ('iou_best', 'iou_normal', 'coords', 'prob')   # dict insertion order :90-94


def objectives(m, labels):
    """model/yolo2/__init__.py:62-94.  labels = (mask[B,cells,1], prob[B,cells,1,C],
    coords[B,cells,1,4], offset_xy_min[B,cells,1,2], offset_xy_max[B,cells,1,2], areas[B,cells,1]).
    Returns (dict of 4 scalars, aux dict with mask_best etc.)."""
    mask, prob, coords, oxy_min, oxy_max, areas = labels
    t = m['iou'].dtype.type
    _min = np.maximum(m['offset_xy_min'], oxy_min)                       # :73
    _max = np.minimum(m['offset_xy_max'], oxy_max)                       # :74
    _wh = np.maximum(_max - _min, t(0))                                  # :75
    _areas = _wh[..., 0] * _wh[..., 1]                                   # :76
    union = np.maximum(areas + m['areas'] - _areas, t(1e-10))            # :77
    iou = _areas / union                                                 # :78
    best_iou = iou.max(2, keepdims=True)                                 # :80
    best_box = (iou == best_iou).astype(iou.dtype)                       # :81 exact equality
    mask_best = mask * best_box                                          # :82
    mask_normal = t(1) - mask_best                                       # :83
    iou_dist = (m['iou'] - mask_best) ** 2                               # :85
    coords_dist = (m['coords'] - coords) ** 2                            # :86
    prob_dist = (m['prob'] - prob) ** 2                                  # :87
    cnt = t(iou_dist.size)                                               # :89 static shape product
    mb = mask_best[..., None]
    obj = {
        'iou_best': (mask_best * iou_dist).sum(dtype=np.float64) / cnt,
        'iou_normal': (mask_normal * iou_dist).sum(dtype=np.float64) / cnt,
        'coords': (mb * coords_dist).sum(dtype=np.float64) / cnt,
        'prob': (mb * prob_dist).sum(dtype=np.float64) / cnt,
    }
    obj = {k: t(v) for k, v in obj.items()}
    return obj, {'mask_best': mask_best, 'iou': iou, 'cnt': cnt}


def total_loss(obj, hparam):
    """train.py:113 tf.losses.get_total_loss = sum of weighted objectives
    (model/yolo2/__init__.py:117-119); yolo2 adds no regularisers."""
    return sum(obj[k] * type(obj[k])(hparam[k]) for k in OBJECTIVE_KEYS)


def loss_backward(m, labels, aux, hparam, classes):
    """d total_loss / d net, derived from model/yolo2/__init__.py:36-49,85-94.  The IoU feeds
    only tf.equal (:81), so no gradient flows through it."""
    mask, prob_t, coords_t, _, _, _ = labels
    t = m['iou'].dtype.type
    mb = aux['mask_best']
    cnt = aux['cnt']
    two = t(2)
    b, cells, a = m['iou'].shape
    dz = np.zeros((b, cells, a, 5 + classes), dtype=m['iou'].dtype)
    s = m['iou']
    w_obj = t(hparam['iou_best']) * mb + t(hparam['iou_normal']) * (t(1) - mb)
    dz[..., 0] = two * (s - mb) * w_obj / cnt * s * (t(1) - s)
    sxy = m['offset_xy']
    dz[..., 1:3] = two * mb[..., None] * (sxy - coords_t[..., :2]) * t(hparam['coords']) / cnt * sxy * (t(1) - sxy)
    sq = m['wh01_sqrt']
    dz[..., 3:5] = two * mb[..., None] * (sq - coords_t[..., 2:4]) * t(hparam['coords']) / cnt * sq / two
    p = m['prob']
    d = two * mb[..., None] * (p - prob_t) * t(hparam['prob']) / cnt
    dz[..., 5:] = p * (d - (d * p).sum(-1, keepdims=True))
    return dz.reshape(b, m['cell_height'], m['cell_width'], a * (5 + classes))


# --------------------------------------------------------------------------------------
# optimizers (train.py:70-80) and learning-rate schedule (train.py:116-124, config.ini:30-33)
# --------------------------------------------------------------------------------------

def exponential_decay(lr, global_step, decay_steps, decay_rate, staircase):
    p = global_step / decay_steps
    if staircase:
        p = np.floor(p)
    return lr * decay_rate ** p


def adam_step(w, g, m, v, lr, t, beta1=0.9, beta2=0.999, eps=1e-8):
    """[TF-sem] tf.train.AdamOptimizer (ApplyAdam kernel): alpha = lr*sqrt(1-b2^t)/(1-b1^t);
    m += (g-m)(1-b1); v += (g^2-v)(1-b2); w -= alpha*m/(sqrt(v)+eps)  (eps outside the sqrt)."""
    ty = w.dtype.type
    alpha = ty(lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t))
    m = m + (g - m) * (ty(1) - ty(beta1))        # (T(1) - beta1) is formed in T by the TF kernel
    v = v + (g * g - v) * (ty(1) - ty(beta2))
    w = w - (m * alpha) / (np.sqrt(v) + ty(eps))
    return w, m, v


def momentum_step(w, g, acc, lr, momentum=0.9):
    """[TF-sem] ApplyMomentum (no nesterov): acc = acc*momentum + g; w -= lr*acc."""
    ty = w.dtype.type
    acc = acc * ty(momentum) + g
    return w - ty(lr) * acc, acc


def gd_step(w, g, lr):
    return w - w.dtype.type(lr) * g


def rmsprop_step(w, g, ms, mom, lr, decay=0.9, momentum=0.0, eps=1e-10):
    """[TF-sem] ApplyRMSProp: ms += (g^2-ms)(1-decay); mom = mom*momentum + lr*g/sqrt(ms+eps); w -= mom."""
    ty = w.dtype.type
    ms = ms + (g * g - ms) * (ty(1) - ty(decay))
    mom = mom * ty(momentum) + ty(lr) * g / np.sqrt(ms + ty(eps))
    return w - mom, ms, mom


def adagrad_step(w, g, acc, lr):
    """[TF-sem] ApplyAdagrad: acc += g^2; w -= lr*g/sqrt(acc)."""
    acc = acc + g * g
    return w - w.dtype.type(lr) * g / np.sqrt(acc), acc


def adadelta_step(w, g, acc, acc_update, lr, rho=0.95, eps=1e-8):
    """[TF-sem] ApplyAdadelta: acc = rho*acc+(1-rho)g^2; upd = sqrt(acc_update+eps)/sqrt(acc+eps)*g;
    acc_update = rho*acc_update+(1-rho)upd^2; w -= lr*upd."""
    ty = w.dtype.type
    acc = acc * ty(rho) + g * g * (ty(1) - ty(rho))
    upd = np.sqrt(acc_update + ty(eps)) / np.sqrt(acc + ty(eps)) * g
    acc_update = acc_update * ty(rho) + upd * upd * (ty(1) - ty(rho))
    return w - ty(lr) * upd, acc, acc_update


def ftrl_step(w, g, accum, linear, lr, lr_power=-0.5, l1=0.0, l2=0.0):
    """[TF-sem] tf.train.FtrlOptimizer (train.py:78; config.ini:55-59) -- ApplyFtrl kernel of TF 1.0:
    new_accum = accum + g^2; linear += g - (new_accum^-p - accum^-p)/lr * w (sqrt for p = -0.5);
    w = (l1*sign(linear) - linear) / (new_accum^-p/lr + 2*l2) where |linear| > l1, else 0; accum = new_accum.
    Slots: accum starts at initial_accumulator_value (0.1), linear at 0."""
    ty = w.dtype.type
    new_accum = accum + g * g
    if lr_power == -0.5:
        pa, pna = np.sqrt(accum), np.sqrt(new_accum)
    else:
        pa, pna = np.power(accum, ty(-lr_power)), np.power(new_accum, ty(-lr_power))
    linear = linear + (g - (pna - pa) / ty(lr) * w)
    x = ty(l1) * np.sign(linear) - linear
    y = pna / ty(lr) + ty(2) * ty(l2)
    w = np.where(np.abs(linear) > ty(l1), x / y, ty(0)).astype(w.dtype)
    return w, new_accum, linear


def clip_by_norm(g, clip):
    """[TF-sem] slim create_train_op clip_gradient_norm -> per-tensor tf.clip_by_norm."""
    n = np.sqrt((g.astype(np.float64) ** 2).sum())
    return g * g.dtype.type(clip / max(n, clip)) if clip > 0 else g


# --------------------------------------------------------------------------------------
# detect-side helpers
# --------------------------------------------------------------------------------------

def per_image_standardization(image):
    """utils/preprocess.py:23-25 (== [TF-sem] tf.image.per_image_standardization, train.py:103):
    (x-mean)/max(std, 1/sqrt(N)), population std over the whole image."""
    stddev = np.std(image)
    return (image - np.mean(image)) / max(stddev, 1.0 / np.sqrt(np.multiply.reduce(image.shape)))


def iou(xy_min1, xy_max1, xy_min2, xy_max2):
    """utils/postprocess.py:21-36, fp32 op order (a1+a2)-inter, floor 1e-10."""
    areas1 = np.multiply.reduce(xy_max1 - xy_min1)
    areas2 = np.multiply.reduce(xy_max2 - xy_min2)
    _xy_min = np.maximum(xy_min1, xy_min2)
    _xy_max = np.minimum(xy_max1, xy_max2)
    _wh = np.maximum(_xy_max - _xy_min, 0)
    _areas = np.multiply.reduce(_wh)
    return _areas / np.maximum(areas1 + areas2 - _areas, 1e-10)


def non_max_suppress(conf, xy_min, xy_max, threshold, threshold_iou):
    """utils/postprocess.py:39-51 restated with explicit indices instead of a list of views.
    Mutates ``conf`` in place; returns ``order`` (box indices in the order the reference's
    returned list has, i.e. after the last class's stable sort)."""
    cells, a, classes = conf.shape
    n = cells * a
    cf = conf.reshape(n, classes)
    mn = xy_min.reshape(n, 2)
    mx = xy_max.reshape(n, 2)
    order = list(range(n))
    for c in range(classes):
        order.sort(key=lambda i: cf[i, c], reverse=True)          # :43 stable, carried order
        for p in range(n - 1):                                     # :44
            i = order[p]
            if cf[i, c] <= threshold:                              # :46-47
                continue
            for j in order[p + 1:]:                                # :48
                if iou(mn[i], mx[i], mn[j], mx[j]) >= threshold_iou:   # :49
                    cf[j, c] = 0                                   # :50
    return np.asarray(order, dtype=np.int64)


def non_max_suppress_fast(conf, xy_min, xy_max, threshold, threshold_iou):
    """Vectorised restatement of the same algorithm (identical fp32 arithmetic per pair, same
    carried stable order); used where the pure-Python loop above is too slow.  Checked against
    :func:`non_max_suppress` and the reference goldens in tests/test_oracle.py."""
    cells, a, classes = conf.shape
    n = cells * a
    cf = conf.reshape(n, classes)
    mn = xy_min.reshape(n, 2).astype(np.float32)
    mx = xy_max.reshape(n, 2).astype(np.float32)
    thr = np.float32(threshold)
    thr_iou = np.float32(threshold_iou)
    wh = mx - mn
    area = wh[:, 0] * wh[:, 1]
    order = np.arange(n)
    for c in range(classes):
        key = cf[order, c]
        order = order[np.argsort(-key, kind='stable')]   # stable descending, ties keep carried order (:43)
        col = cf[:, c]
        for p in range(n - 1):
            i = order[p]
            if col[i] <= thr:
                continue
            rest = order[p + 1:]
            _min = np.maximum(mn[i], mn[rest])
            _max = np.minimum(mx[i], mx[rest])
            _wh = np.maximum(_max - _min, np.float32(0))
            inter = _wh[:, 0] * _wh[:, 1]
            v = inter / np.maximum((area[i] + area[rest]) - inter, np.float32(1e-10))
            col[rest[v >= thr_iou]] = 0
    return order.astype(np.int64)


def transform_labels(objects_class, objects_coord, classes, cell_width, cell_height, dtype=np.float32):
    """utils/data/__init__.py:112-145.  coords normalised (xmin,ymin,xmax,ymax) in [0,1]."""
    cells = cell_height * cell_width
    mask = np.zeros([cells, 1], dtype=dtype)
    prob = np.zeros([cells, 1, classes], dtype=dtype)
    coords = np.zeros([cells, 1, 4], dtype=dtype)
    offset_xy_min = np.zeros([cells, 1, 2], dtype=dtype)
    offset_xy_max = np.zeros([cells, 1, 2], dtype=dtype)
    objects_class = np.asarray(objects_class)
    objects_coord = np.asarray(objects_coord)
    if len(objects_class):
        xmin, ymin, xmax, ymax = objects_coord.T
        x = cell_width * (xmin + xmax) / 2                         # :121
        y = cell_height * (ymin + ymax) / 2
        ix, iy = np.floor(x), np.floor(y)
        ox, oy = x - ix, y - iy
        w, h = xmax - xmin, ymax - ymin
        index = (iy * cell_width + ix).astype(int)                 # :129
        mask[index, :] = 1
        prob[index, :, objects_class] = 1                          # multi-hot if a cell is shared
        coords[index, 0, 0] = ox
        coords[index, 0, 1] = oy
        coords[index, 0, 2] = np.sqrt(w)
        coords[index, 0, 3] = np.sqrt(h)
        _w, _h = w / 2 * cell_width, h / 2 * cell_height
        offset_xy_min[index, 0, 0] = ox - _w
        offset_xy_min[index, 0, 1] = oy - _h
        offset_xy_max[index, 0, 0] = ox + _w
        offset_xy_max[index, 0, 1] = oy + _h
    wh = offset_xy_max - offset_xy_min
    areas = np.multiply.reduce(wh, -1)
    return mask, prob, coords, offset_xy_min, offset_xy_max, areas


# --------------------------------------------------------------------------------------
# one full training step / detect pass (callers train.py:109-129, detect.py:69-71)
# --------------------------------------------------------------------------------------

def train_step(spec, params, opt_state, x, labels, classes, anchors, hparam, lr, step,
               adam=(0.9, 0.999, 1e-8), quant=None):
    """Forward (batch-stat BN) + loss + backward + Adam; returns (new_params, new_state, info).
    ``step`` counts completed updates (Adam's t = step+1)."""
    net, caches = network_forward(spec, params, x, training=True, quant=quant)
    m = model_decode(net, classes, anchors, training=True)
    obj, aux = objectives(m, labels)
    loss = total_loss(obj, hparam)
    dnet = loss_backward(m, labels, aux, hparam, classes)
    grads = network_backward(spec, params, caches, dnet, quant=quant)
    new_params = dict(params)
    new_params.update(caches['ema'])                     # UPDATE_OPS before the step [TF-sem]
    new_state = {}
    for k in trainable_names(params):
        mm, vv = opt_state.get(k, (np.zeros_like(params[k]), np.zeros_like(params[k])))
        w, mm, vv = adam_step(params[k], grads[k], mm, vv, lr, step + 1, *adam)
        new_params[k] = w
        new_state[k] = (mm, vv)
    return new_params, new_state, {'loss': loss, 'objectives': obj, 'grads': grads, 'net': net}


def detect(spec, params, x, classes, anchors, threshold=0.3, threshold_iou=0.4):
    """detect.py:69-80 for a batch: forward (moving-stat BN) -> decode -> per-image NMS."""
    net, _ = network_forward(spec, params, x, training=False)
    m = model_decode(net, classes, anchors, training=False)
    conf = m['conf'].copy()
    orders = []
    for b in range(conf.shape[0]):
        orders.append(non_max_suppress_fast(conf[b], m['xy_min'][b], m['xy_max'][b], threshold, threshold_iou))
    return conf, m['xy_min'], m['xy_max'], orders


# ---------------------------------------------------------------------------------------------------------
# Input pipeline: utils/data/__init__.py:50-109,162-175 and utils/preprocess.py:28-71.
# The image arithmetic lives in TensorFlow 1.0 (tf.image.*), which is not under /root/reference and cannot run
# here: PARITY UNPINNED for these functions -- they restate the published TF-1.0 kernels ([TF-sem]:
# ResizeBilinear align_corners=False, RGBToHSV / HSVToRGB, AdjustContrastv2, rgb_to_grayscale weights) in float32
# in the kernels' operation order.  The box arithmetic (random_crop, flip, resize factor) is the reference's own.
# ---------------------------------------------------------------------------------------------------------
def random_crop_box(objects_coord, width_height, u4, scale):
    """utils/preprocess.py:28-42 given the four uniforms u4 ~ U(0, scale) it draws.  objects_coord [K,4] pixels.
    Returns (coords shifted into the crop, integer crop (x0, y0, w, h) for crop_to_bounding_box, float _wh)."""
    f = np.float32
    coord = np.asarray(objects_coord, f)
    wh = np.asarray(width_height, f)
    xy_min = coord[:, :2].min(0)
    xy_max = coord[:, 2:].max(0)
    margin = wh - xy_max
    shrink = np.asarray(u4, f) * np.concatenate([xy_min, margin]).astype(f)
    _xy_min = shrink[:2]
    _wh = wh - shrink[2:] - _xy_min
    coord = coord - np.tile(_xy_min, 2)
    x0, y0 = int(_xy_min[0]), int(_xy_min[1])          # tf.cast(float -> int32) truncates
    w, h = int(_wh[0]), int(_wh[1])
    return coord, (x0, y0, w, h), _wh


def resize_bilinear(image, out_h, out_w):
    """tf.image.resize_images(image, [h, w]) of TF 1.0 = ResizeBilinear, align_corners=False: in = out * (in/out) (no
    half-pixel offset), top-left / bottom-right taps, f32.  Same size -> returned unchanged (resize_images' shortcut)."""
    f = np.float32
    img = np.asarray(image, f)
    in_h, in_w = img.shape[:2]
    if (in_h, in_w) == (out_h, out_w):
        return img
    hs, ws = f(in_h) / f(out_h), f(in_w) / f(out_w)
    ys = np.arange(out_h, dtype=f) * hs
    xs = np.arange(out_w, dtype=f) * ws
    y0 = np.floor(ys).astype(np.int64)
    x0 = np.floor(xs).astype(np.int64)
    y1 = np.minimum(np.ceil(ys).astype(np.int64), in_h - 1)
    x1 = np.minimum(np.ceil(xs).astype(np.int64), in_w - 1)
    ly = (ys - y0.astype(f))[:, None, None]
    lx = (xs - x0.astype(f))[None, :, None]
    tl, tr = img[y0][:, x0], img[y0][:, x1]
    bl, br = img[y1][:, x0], img[y1][:, x1]
    top = tl + (tr - tl) * lx
    bottom = bl + (br - bl) * lx
    return (top + (bottom - top) * ly).astype(f)


def resize_coords(objects_coord, crop_wh, width, height):
    """utils/data/__init__.py:63-68: factor = [width, height] / width_height (the FLOAT extent random_crop returned)."""
    f = np.float32
    factor = np.array([width, height], f) / np.asarray(crop_wh, f)
    return np.asarray(objects_coord, f) * np.tile(factor, 2)


def flip_coords(objects_coord, width):
    """utils/preprocess.py:45-51."""
    c = np.asarray(objects_coord, np.float32)
    w = np.float32(width)
    return np.stack([w - c[:, 2], c[:, 1], w - c[:, 0], c[:, 3]], 1)


def rgb_to_hsv(rgb):
    """TF RGBToHSV kernel, f32 (scale-free in V)."""
    f = np.float32
    r, g, b = [np.asarray(rgb[..., i], f) for i in range(3)]
    v = np.maximum(np.maximum(r, g), b)
    rng = v - np.minimum(np.minimum(r, g), b)
    with np.errstate(divide='ignore', invalid='ignore'):
        s = np.where(v > 0, rng / v, f(0)).astype(f)
        norm = (f(1) / (f(6) * rng)).astype(f)
        h = np.where(r == v, norm * (g - b), np.where(g == v, norm * (b - r) + f(2.0 / 6.0), norm * (r - g) + f(4.0 / 6.0))).astype(f)
    h = np.where(rng <= 0, f(0), h)
    h = np.where(h < 0, h + f(1), h).astype(f)
    return np.stack([h, s, v], -1)


def hsv_to_rgb(hsv):
    """TF HSVToRGB kernel, f32."""
    f = np.float32
    h, s, v = [np.asarray(hsv[..., i], f) for i in range(3)]
    c = s * v
    m = v - c
    dh = h * f(6)
    fmodu = dh.copy()
    fmodu = np.where(fmodu <= 0, fmodu + f(2) * np.ceil((-fmodu) / f(2) + (fmodu == np.floor(fmodu / 2) * 2)), fmodu)   # while (<= 0) += 2
    fmodu = (fmodu - f(2) * np.floor(fmodu / f(2))).astype(f)            # while (>= 2) -= 2
    x = c * (f(1) - np.abs(fmodu - f(1)))
    cat = dh.astype(np.int32)
    z = np.zeros_like(c)
    rr = np.select([cat == 0, cat == 1, cat == 4, cat == 5], [c, x, x, c], z)
    gg = np.select([cat == 0, cat == 1, cat == 2, cat == 3], [x, c, c, x], z)
    bb = np.select([cat == 2, cat == 3, cat == 4, cat == 5], [x, c, c, x], z)
    return np.stack([rr + m, gg + m, bb + m], -1).astype(f)


def adjust_saturation(image, factor):
    hsv = rgb_to_hsv(image)
    hsv[..., 1] = np.clip(hsv[..., 1] * np.float32(factor), 0, 1)
    return hsv_to_rgb(hsv)


def adjust_hue(image, delta):
    hsv = rgb_to_hsv(image)
    hsv[..., 0] = np.mod(hsv[..., 0] + (np.float32(1) + np.float32(delta)), np.float32(1)).astype(np.float32)
    return hsv_to_rgb(hsv)


def adjust_contrast(image, factor):
    """AdjustContrastv2: per-channel mean over H x W."""
    img = np.asarray(image, np.float32)
    mean = img.mean((0, 1), dtype=np.float64).astype(np.float32)
    return ((img - mean) * np.float32(factor) + mean).astype(np.float32)


def rgb_to_grayscale3(image):
    """tf.image.rgb_to_grayscale weights, tiled back to 3 channels (utils/preprocess.py:63-71)."""
    img = np.asarray(image, np.float32)
    g = img[..., 0] * np.float32(0.2989) + img[..., 1] * np.float32(0.5870) + img[..., 2] * np.float32(0.1140)
    return np.repeat(g[..., None], 3, -1).astype(np.float32)


def augment_image(src_u8, p, width, height):
    """load_image_labels image path (utils/data/__init__.py:162-172) for ONE image with every random draw supplied in
    the dict p: crop (x0,y0,w,h) or None, flip, brightness / saturation / hue / contrast (None = branch not taken),
    noise (None or an [H,W,3] array already scaled), gray.  Returns f32 [height, width, 3] in 0..255."""
    img = np.asarray(src_u8, np.float32)
    if p.get('crop') is not None:
        x0, y0, w, h = p['crop']
        img = img[y0:y0 + h, x0:x0 + w]
    img = resize_bilinear(img, height, width)
    if p.get('flip'):
        img = img[:, ::-1]
    if p.get('brightness') is not None:
        img = img + np.float32(p['brightness'])
    if p.get('saturation') is not None:
        img = adjust_saturation(img, p['saturation'])
    if p.get('hue') is not None:
        img = adjust_hue(img, p['hue'])
    if p.get('contrast') is not None:
        img = adjust_contrast(img, p['contrast'])
    if p.get('noise') is not None:
        img = img + np.asarray(p['noise'], np.float32)
    if p.get('gray'):
        img = rgb_to_grayscale3(img)
    return np.clip(img, 0, 255).astype(np.float32)


# ---------------------------------------------------------------------------------------------------------
# YOLO (v1) family: model/yolo/inference.py:24-66 (tiny) and model/yolo/__init__.py:37-100.
# [TF-sem] slim.layers.conv2d / fully_connected without a normalizer: weights (Xavier uniform) + biases (zeros) + activation;
# slim.layers.flatten on NHWC = row-major (h, w, c); slim.layers.dropout(keep_prob) = x * mask / keep_prob;
# slim.l2_regularizer(s)(w) = s * sum(w^2) / 2, added to tf.losses.get_total_loss.  PARITY UNPINNED (TensorFlow), as for yolo2.
# ---------------------------------------------------------------------------------------------------------

def yolo1_tiny_spec(classes, boxes_per_cell, cells):
    ops = []
    for i, (ch, pooled) in enumerate(((16, 1), (32, 1), (64, 1), (128, 1), (256, 1), (512, 1), (512, 0), (1024, 0), (256, 0))):
        ops.append(('convb', 'conv%d' % i, 3, ch))
        if pooled:
            ops.append(('pool', 2))
    ops.append(('flatten',))
    for i, units in enumerate((256, 4096)):
        ops += [('fc', 'fc%d' % i, units, True, 0.001), ('dropout', 'dropout%d' % i, 0.5)]
    ops.append(('fc', 'fc', cells * (classes + boxes_per_cell * 5), False, 0.0))     # outside the regularised arg_scope (model/yolo/inference.py:62)
    return ops


def yolo1_forward(spec, params, x, masks=None, quant=None):
    """masks: {dropout name: 0/1 array} (training) or None (inference: dropout is the identity).  Returns (net [B, N], caches)."""
    q = quant or _ident
    net = q(x)
    caches = []
    for op in spec:
        if op[0] == 'convb':
            _, name, k, cout = op
            out = q(leaky_relu(conv2d(net, q(params[name + '/weights'])) + params[name + '/biases']))
            caches.append(('convb', name, net, out))
            net = out
        elif op[0] == 'pool':
            caches.append(('pool', op[1], net))
            net = max_pool(net, op[1])
        elif op[0] == 'flatten':
            caches.append(('flatten', net.shape))
            net = net.reshape(net.shape[0], -1)
        elif op[0] == 'fc':
            _, name, cout, act, l2 = op
            z = net @ q(params[name + '/weights']) + params[name + '/biases']
            out = q(leaky_relu(z) if act else z)
            caches.append(('fc', name, net, out))
            net = out
        elif op[0] == 'dropout':
            if masks is not None:
                m = masks[op[1]].reshape(net.shape).astype(net.dtype)
                caches.append(('dropout', m, op[2]))
                net = q(net * m / net.dtype.type(op[2]))
            else:
                caches.append(('dropout', None, op[2]))
    return net, caches


def yolo1_backward(spec, params, caches, dnet, quant=None):
    q = quant or _ident
    grads = {}
    reg = 0.0
    dnet = q(dnet)
    for op, cache in zip(reversed(spec), reversed(caches)):
        if op[0] == 'fc':
            _, name, cout, act, l2 = op
            _, _, xin, out = cache
            dz = q(leaky_relu_grad(out, dnet)) if act else dnet
            w = params[name + '/weights']
            grads[name + '/weights'] = xin.T @ dz + w.dtype.type(l2) * w
            grads[name + '/biases'] = dz.astype(np.float64).sum(0).astype(dz.dtype)
            reg += l2 * float((w.astype(np.float64) ** 2).sum()) / 2
            dnet = q(dz @ q(w).T)
        elif op[0] == 'dropout':
            if cache[1] is not None:
                dnet = q(dnet * cache[1] / dnet.dtype.type(cache[2]))
        elif op[0] == 'flatten':
            dnet = dnet.reshape(cache[1])
        elif op[0] == 'pool':
            dnet = max_pool_grad(cache[2], dnet, cache[1])
        elif op[0] == 'convb':
            _, name, k, cout = op
            _, _, xin, out = cache
            dz = q(leaky_relu_grad(out, dnet))             # sign(out) == sign(z) for alpha > 0
            grads[name + '/weights'] = conv2d_wgrad(xin, dz, k, k)
            grads[name + '/biases'] = dz.reshape(-1, cout).astype(np.float64).sum(0).astype(dz.dtype)
            dnet = q(conv2d_dgrad(dz, q(params[name + '/weights'])))
    return grads, reg


def yolo1_model_decode(net, classes, boxes_per_cell, cell_height, cell_width, training=False):
    """model/yolo/__init__.py:37-66."""
    t = net.dtype.type
    b = net.shape[0]
    cells = cell_height * cell_width
    end = cells * classes
    m = {'cell_width': cell_width, 'cell_height': cell_height}
    m['prob'] = net[:, :end].reshape(b, cells, 1, classes)                                    # :43
    rem = net[:, end:end + cells * boxes_per_cell * 5].reshape(b, cells, boxes_per_cell, 5)   # :44
    m['iou'] = rem[..., 0]
    m['offset_xy'] = rem[..., 1:3]
    base = rem[..., 3:]
    m['wh01_sqrt_base'] = base
    wh01 = base * base                                                                        # :48
    m['coords'] = np.concatenate([m['offset_xy'], np.abs(base)], -1)                          # :49-50
    m['wh'] = wh01 * np.array([cell_width, cell_height], net.dtype)                           # :51
    half = m['wh'] / t(2)
    m['offset_xy_min'] = m['offset_xy'] - half
    m['offset_xy_max'] = m['offset_xy'] + half
    m['areas'] = m['wh'][..., 0] * m['wh'][..., 1]
    if not training:
        cell_xy = calc_cell_xy(cell_height, cell_width, net.dtype).reshape(1, cells, 1, 2)
        m['xy'] = cell_xy + m['offset_xy']
        m['xy_min'] = cell_xy + m['offset_xy_min']
        m['xy_max'] = cell_xy + m['offset_xy_max']
        m['conf'] = m['iou'][..., None] * m['prob']
    return m


def yolo1_objectives(m, labels):
    """model/yolo/__init__.py:69-100: as yolo2's, except that the class term is per cell and masked by `mask` (:100)."""
    mask, prob, coords, oxy_min, oxy_max, areas = labels
    t = m['iou'].dtype.type
    _wh = np.maximum(np.minimum(m['offset_xy_max'], oxy_max) - np.maximum(m['offset_xy_min'], oxy_min), t(0))
    _areas = _wh[..., 0] * _wh[..., 1]
    iou = _areas / np.maximum(areas + m['areas'] - _areas, t(1e-10))
    mask_best = mask * (iou == iou.max(2, keepdims=True)).astype(iou.dtype)
    iou_dist = (m['iou'] - mask_best) ** 2
    coords_dist = (m['coords'] - coords) ** 2
    prob_dist = (m['prob'] - prob) ** 2
    cnt = t(iou_dist.size)
    obj = {'iou_best': (mask_best * iou_dist).sum(dtype=np.float64) / cnt, 'iou_normal': ((t(1) - mask_best) * iou_dist).sum(dtype=np.float64) / cnt,
           'coords': (mask_best[..., None] * coords_dist).sum(dtype=np.float64) / cnt, 'prob': (mask[..., None] * prob_dist).sum(dtype=np.float64) / cnt}
    return {k: t(v) for k, v in obj.items()}, {'mask_best': mask_best, 'cnt': cnt}


def yolo1_loss_backward(m, labels, aux, hparam, classes, boxes_per_cell, width):
    """d(sum of weighted objectives)/d(net); every output is linear, |x| differentiates to sign(x)."""
    mask, prob_t, coords_t, _, _, _ = labels
    t = m['iou'].dtype.type
    mb, cnt, two = aux['mask_best'], aux['cnt'], t(2)
    b, cells, _ = m['iou'].shape
    d = np.zeros((b, width), m['iou'].dtype)
    d[:, :cells * classes] = (two * mask[..., None] * (m['prob'] - prob_t) * t(hparam['prob']) / cnt).reshape(b, -1)
    r = np.zeros((b, cells, boxes_per_cell, 5), m['iou'].dtype)
    r[..., 0] = two * (m['iou'] - mb) * (t(hparam['iou_best']) * mb + t(hparam['iou_normal']) * (t(1) - mb)) / cnt
    r[..., 1:3] = two * mb[..., None] * (m['offset_xy'] - coords_t[..., :2]) * t(hparam['coords']) / cnt
    base = m['wh01_sqrt_base']
    r[..., 3:5] = two * mb[..., None] * (np.abs(base) - coords_t[..., 2:4]) * t(hparam['coords']) / cnt * np.sign(base)
    d[:, cells * classes:cells * (classes + boxes_per_cell * 5)] = r.reshape(b, -1)
    return d

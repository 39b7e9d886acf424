"""CPU BASELINE LEG -- test infrastructure only (same rules as oracle/yolo2_ref.py: imported by tests/ and by
bench.py's ``cpu_baseline`` leg, never by the product).

SURVEY 8(d) / BASELINE.md section 3 ask for "the reference's TF1 CPU path" timed beside the GPU numbers.  TensorFlow 1.0
cannot run in this image (not installed, no network), so the closest honest stand-in for what TF-1.0's Eigen/MKL CPU
kernels do is the same network evaluated by torch-CPU (oneDNN convolutions, NHWC/channels_last, fp32) at the reference's
default batch size 8 (train.py:156): conv stack forward, and forward + backward.  The topology comes from
``yolo2_ref.darknet_spec`` (a restatement of model/yolo2/inference.py:61-120); batch norm uses batch statistics
(training), leaky ReLU 0.1, 2x2 max pools, reorg + concat.  The loss is the NumPy oracle's (microseconds at this size);
for the backward timing the network output is reduced with a fixed random cotangent, which has the same cost.
"""
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import yolo2_ref as R


def build(spec, params):
    """-> list of (op, tensors) with torch parameters (requires_grad) in NCHW/channels_last."""
    layers = []
    for op in spec:
        if op[0] == 'conv':
            _, name, k, cout, bn = op
            w = torch.from_numpy(np.ascontiguousarray(params[name + '/weights'].transpose(3, 2, 0, 1))).contiguous(memory_format=torch.channels_last)
            w.requires_grad_(True)
            if bn:
                g = torch.from_numpy(params[name + '/BatchNorm/gamma'].copy()).requires_grad_(True)
                b = torch.from_numpy(params[name + '/BatchNorm/beta'].copy()).requires_grad_(True)
                layers.append(('conv_bn', k, w, g, b))
            else:
                b = torch.from_numpy(params[name + '/biases'].copy()).requires_grad_(True)
                layers.append(('conv_bias', k, w, b))
        else:
            layers.append(op)
    return layers


def forward(layers, x_nhwc, training=True):
    """x_nhwc: torch f32 [B,H,W,3]; returns the network output NHWC."""
    t = x_nhwc.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
    mark = None
    for layer in layers:
        kind = layer[0]
        if kind == 'conv_bn':
            _, k, w, g, b = layer
            t = F.conv2d(t, w, None, 1, k // 2)
            t = F.batch_norm(t, None, None, g, b, True, 0.0, R.BN_EPS)
            t = torch.maximum(t, 0.1 * t)
        elif kind == 'conv_bias':
            _, k, w, b = layer
            t = F.conv2d(t, w, b, 1, k // 2)
        elif kind == 'pool':
            if layer[1] == 2:
                t = F.max_pool2d(t, 2, 2)
            else:
                t = F.max_pool2d(F.pad(t, (0, 1, 0, 1), value=float('-inf')), 2, 1)
        elif kind == 'mark':
            mark = t
        elif kind == 'reorg_concat':
            b_, c, h, w_ = mark.shape
            r = mark.reshape(b_, c, h // 2, 2, w_ // 2, 2).permute(0, 3, 5, 1, 2, 4).reshape(b_, 4 * c, h // 2, w_ // 2)
            t = torch.cat([r, t], 1)
    return t.permute(0, 2, 3, 1)


def time_conv_stack(classes=20, num_anchors=5, size=416, batch=8, budget_s=12.0, threads=None, seed=0):
    """-> dict(fwd_img_s, train_img_s, threads, iters).  One warm-up pass of each kind, then as many timed passes as fit."""
    if threads:
        torch.set_num_threads(int(threads))
    spec = R.darknet_spec(classes, num_anchors)
    layers = build(spec, R.init_params(spec, seed=seed))
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, size, size, 3, generator=g)
    cot = torch.randn(batch, size // 32, size // 32, num_anchors * (5 + classes), generator=g)

    def fwd():
        with torch.no_grad():
            forward(layers, x)

    def train():
        out = forward(layers, x)
        (out * cot).sum().backward()

    res = {'threads': torch.get_num_threads(), 'batch': batch}
    for key, fn in (('fwd', fwd), ('train', train)):
        fn()
        n, t0 = 0, time.time()
        while n < 1 or (time.time() - t0) < budget_s / 2:
            fn()
            n += 1
        dt = time.time() - t0
        res[key + '_img_s'] = batch * n / dt
        res[key + '_iters'] = n
    return res

"""Pins the CPU oracle (oracle/yolo2_ref.py, oracle/nms_ref.c): against the golden vectors the
reference itself produced (tests/golden/make_golden.py), against torch-CPU fp64 autograd for the
TF-semantics parts the reference cannot pin, and against analytic known answers."""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import yolo2_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def nms_g(golden_dir):
    return np.load(os.path.join(golden_dir, 'nms.npz'))


@pytest.fixture(scope='module')
def lab_g(golden_dir):
    return np.load(os.path.join(golden_dir, 'labels.npz'))


NMS_CASES = ['sparse20', 'sparse80', 'dense', 'dense845', 'clustered', 'identical_ties', 'ties', 'at_threshold', 'all_below', 'zero_area']


def _nms_lib():
    path = os.path.join(ROOT, 'oracle', 'libnms_ref.so')
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')])
    lib = ctypes.CDLL(path)
    lib.nms_ref.restype = ctypes.c_int
    return lib


def c_nms(conf, mn, mx, thr, thr_iou):
    lib = _nms_lib()
    n, c = conf.shape[0] * conf.shape[1], conf.shape[2]
    conf = np.ascontiguousarray(conf, np.float32)
    mn = np.ascontiguousarray(mn, np.float32)
    mx = np.ascontiguousarray(mx, np.float32)
    order = np.zeros(n, np.int64)
    P = ctypes.POINTER(ctypes.c_float)
    rc = lib.nms_ref(conf.ctypes.data_as(P), mn.ctypes.data_as(P), mx.ctypes.data_as(P), ctypes.c_long(n), ctypes.c_long(c),
                     ctypes.c_float(thr), ctypes.c_float(thr_iou), order.ctypes.data_as(ctypes.POINTER(ctypes.c_long)))
    assert rc == 0
    return conf, order


@pytest.mark.parametrize('case', NMS_CASES)
def test_nms_fast_and_c_match_reference_golden(nms_g, case):
    conf = nms_g[case + '/conf_in'].copy()
    order = R.non_max_suppress_fast(conf, nms_g[case + '/xy_min'], nms_g[case + '/xy_max'], float(nms_g[case + '/thr']), float(nms_g[case + '/thr_iou']))
    assert np.array_equal(conf, nms_g[case + '/conf_out'])
    assert np.array_equal(order, nms_g[case + '/order'])
    conf_c, order_c = c_nms(nms_g[case + '/conf_in'], nms_g[case + '/xy_min'], nms_g[case + '/xy_max'], float(nms_g[case + '/thr']), float(nms_g[case + '/thr_iou']))
    assert np.array_equal(conf_c, nms_g[case + '/conf_out'])
    assert np.array_equal(order_c, nms_g[case + '/order'])


@pytest.mark.parametrize('case', ['dense', 'identical_ties', 'at_threshold', 'zero_area'])
def test_nms_python_loop_matches_reference_golden(nms_g, case):
    conf = nms_g[case + '/conf_in'].copy()
    order = R.non_max_suppress(conf, nms_g[case + '/xy_min'], nms_g[case + '/xy_max'], float(nms_g[case + '/thr']), float(nms_g[case + '/thr_iou']))
    assert np.array_equal(conf, nms_g[case + '/conf_out'])
    assert np.array_equal(order, nms_g[case + '/order'])


def test_iou_golden(nms_g):
    out = np.array([R.iou(nms_g['iou/a_min'][i], nms_g['iou/a_max'][i], nms_g['iou/b_min'][i], nms_g['iou/b_max'][i]) for i in range(64)], np.float32)
    assert np.array_equal(out, nms_g['iou/out'])


@pytest.mark.parametrize('case', ['voc13', 'coco13', 'rect', 'shared_cell'])
def test_transform_labels_golden(lab_g, case):
    classes, cw, ch = lab_g[case + '/dims']
    out = R.transform_labels(lab_g[case + '/class'], lab_g[case + '/coord'], int(classes), int(cw), int(ch))
    for key, v in zip(('mask', 'prob', 'coords', 'offset_xy_min', 'offset_xy_max', 'areas'), out):
        assert np.array_equal(v, lab_g[case + '/' + key]), key
    # label invariants the reference asserts (utils/visualize.py:46-48)
    mask, _, coords, mn, mx, _ = out
    wh = mx - mn
    assert np.all(wh >= 0)
    sel = mask[:, 0] > 0
    np.testing.assert_allclose(wh[sel, 0] / [cw, ch], coords[sel, 0, 2:4] ** 2, rtol=1e-3)
    np.testing.assert_allclose(mn[sel, 0] + wh[sel, 0] / 2, coords[sel, 0, 0:2], rtol=1e-3, atol=1e-6)


def test_cell_xy_and_standardization_golden(lab_g):
    assert np.array_equal(R.calc_cell_xy(13, 13), lab_g['cell_xy/13x13'])
    assert np.array_equal(R.calc_cell_xy(10, 19), lab_g['cell_xy/10x19'])
    assert np.array_equal(R.per_image_standardization(lab_g['std/in']), lab_g['std/out'])
    assert np.array_equal(R.per_image_standardization(lab_g['std/flat_in']), lab_g['std/flat_out'])


def test_reorg_reference_kat(lab_g):
    out = R.reorg(lab_g['reorg/kat_in'])
    for i, ch in enumerate(np.transpose(out[0], [2, 0, 1])):
        assert np.unique(ch).tolist() == [int(lab_g['reorg/kat_channel_values'][i])]
    # closed form + inverse
    rng = np.random.RandomState(0)
    x = rng.randn(2, 6, 8, 5).astype(np.float32)
    r = R.reorg(x)
    for sy in range(2):
        for sx in range(2):
            assert np.array_equal(r[..., (sy * 2 + sx) * 5:(sy * 2 + sx + 1) * 5], x[:, sy::2, sx::2, :])
    assert np.array_equal(R.reorg_grad(r), x)


# ---- TF-semantics parts: second opinion from torch CPU (fp64) ---------------------------------

def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


@pytest.mark.parametrize('k', [1, 3])
def test_conv_matches_torch(k):
    rng = np.random.RandomState(k)
    x = rng.randn(2, 7, 9, 5)
    w = rng.randn(k, k, 5, 6)
    dy = rng.randn(2, 7, 9, 6)
    xt = _t(x).permute(0, 3, 1, 2).requires_grad_(True)
    wt = _t(w).permute(3, 2, 0, 1).requires_grad_(True)
    yt = F.conv2d(xt, wt, padding=k // 2)
    yt.backward(_t(dy).permute(0, 3, 1, 2))
    np.testing.assert_allclose(R.conv2d(x, w), yt.detach().permute(0, 2, 3, 1).numpy(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(R.conv2d_dgrad(dy, w), xt.grad.permute(0, 2, 3, 1).numpy(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(R.conv2d_wgrad(x, dy, k, k), wt.grad.permute(2, 3, 1, 0).numpy(), rtol=1e-10, atol=1e-10)


def test_conv_delta_kernel_is_identity_and_1x1_is_matmul():
    rng = np.random.RandomState(0)
    x = rng.randn(1, 5, 5, 3).astype(np.float32)
    w = np.zeros((3, 3, 3, 3), np.float32)
    w[1, 1] = np.eye(3)
    assert np.array_equal(R.conv2d(x, w), x)
    w1 = rng.randn(1, 1, 3, 4).astype(np.float32)
    np.testing.assert_allclose(R.conv2d(x, w1), x @ w1[0, 0], rtol=1e-6)


def test_bn_pool_leaky_match_torch():
    rng = np.random.RandomState(1)
    x = rng.randn(3, 6, 6, 4)
    g, b = rng.rand(4) + 0.5, rng.randn(4)
    dz = rng.randn(3, 6, 6, 4)
    xt = _t(x).permute(0, 3, 1, 2).requires_grad_(True)
    gt, bt = _t(g).requires_grad_(True), _t(b).requires_grad_(True)
    zt = F.batch_norm(xt, None, None, gt, bt, training=True, eps=1e-5)
    at = torch.maximum(zt, 0.1 * zt)
    pt = F.max_pool2d(at, 2)
    pt.backward(_t(dz[:, :3, :3, :]).permute(0, 3, 1, 2))
    mean, var = R.bn_moments(x)
    z = R.bn_apply(x, mean, var, g, b)
    a = R.leaky_relu(z)
    p = R.max_pool(a)
    np.testing.assert_allclose(p, pt.detach().permute(0, 2, 3, 1).numpy(), rtol=1e-10)
    da = R.max_pool_grad(a, dz[:, :3, :3, :])
    dzz = R.leaky_relu_grad(z, da)
    dx, dg, db = R.bn_train_bwd(x, mean, var, g, dzz)
    np.testing.assert_allclose(dx, xt.grad.permute(0, 2, 3, 1).numpy(), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(dg, gt.grad.numpy(), rtol=1e-9)
    np.testing.assert_allclose(db, bt.grad.numpy(), rtol=1e-9)
    # constant input -> beta; stride-1 SAME pool (tiny) pads bottom/right only
    c = np.full((2, 4, 4, 3), 2.5)
    m, v = R.bn_moments(c)
    np.testing.assert_allclose(R.bn_apply(c, m, v, np.ones(3), np.array([1., 2., 3.])), np.broadcast_to([1., 2., 3.], c.shape))
    s1 = R.max_pool(x, 1)
    ref = F.max_pool2d(F.pad(_t(x).permute(0, 3, 1, 2), (0, 1, 0, 1), value=-np.inf), 2, stride=1).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(s1, ref)
    assert R.max_pool_grad(x, np.ones_like(x), 1).sum() == x.size


def _torch_loss(net_t, labels, classes, anchors, hp):
    """Independent torch restatement of model/yolo2/__init__.py:28-94 for autograd."""
    b, ch, cw, _ = net_t.shape
    a = len(anchors)
    inp = net_t.reshape(b, ch * cw, a, 5 + classes)
    sig = torch.sigmoid(inp[..., :3])
    iou_p, oxy = sig[..., 0], sig[..., 1:3]
    wh = torch.exp(inp[..., 3:5]) * torch.tensor(anchors, dtype=net_t.dtype).reshape(1, 1, a, 2)
    prob = torch.softmax(inp[..., 5:], -1)
    areas = wh[..., 0] * wh[..., 1]
    mn, mx = oxy - wh / 2, oxy + wh / 2
    coords = torch.cat([oxy, torch.sqrt(wh / torch.tensor([cw, ch], dtype=net_t.dtype))], -1)
    mask, tprob, tcoords, tmn, tmx, tareas = [_t(l) for l in labels]
    iw = torch.clamp(torch.minimum(mx, tmx) - torch.maximum(mn, tmn), min=0)
    inter = iw[..., 0] * iw[..., 1]
    iou = inter / torch.clamp(tareas + areas - inter, min=1e-10)
    best = (iou == iou.max(2, keepdim=True).values).to(net_t.dtype)
    mb = (mask * best).detach()
    cnt = float(mb.numel())
    d_iou = (iou_p - mb) ** 2
    obj = {'iou_best': (mb * d_iou).sum() / cnt, 'iou_normal': ((1 - mb) * d_iou).sum() / cnt,
           'coords': (mb[..., None] * (coords - tcoords) ** 2).sum() / cnt,
           'prob': (mb[..., None] * (prob - tprob) ** 2).sum() / cnt}
    return sum(obj[k] * hp[k] for k in obj), obj


def make_labels(rng, b, classes, cw, ch, dtype=np.float64, kmax=6):
    outs = []
    for _ in range(b):
        k = rng.randint(1, kmax + 1)
        cen = rng.uniform(0.05, 0.95, (k, 2))
        wh = rng.uniform(0.05, 0.6, (k, 2))
        coord = np.clip(np.concatenate([cen - wh / 2, cen + wh / 2], 1), 0, 1)
        outs.append(R.transform_labels(rng.randint(0, classes, k), coord, classes, cw, ch, dtype=dtype))
    return tuple(np.stack([o[i] for o in outs]) for i in range(6))


def test_loss_forward_backward_matches_torch_autograd():
    rng = np.random.RandomState(2)
    classes, anchors = 4, np.array([[1.08, 1.19], [3.42, 4.41], [6.63, 11.38]])
    hp = {'prob': 1., 'iou_best': 5., 'iou_normal': 1., 'coords': 1.}
    net = rng.randn(2, 3, 4, 3 * 9) * 0.5
    labels = make_labels(rng, 2, classes, 4, 3)
    m = R.model_decode(net, classes, anchors, training=True)
    obj, aux = R.objectives(m, labels)
    dnet = R.loss_backward(m, labels, aux, hp, classes)
    nt = _t(net).requires_grad_(True)
    tl, tobj = _torch_loss(nt, labels, classes, anchors, hp)
    tl.backward()
    for k in obj:
        np.testing.assert_allclose(obj[k], tobj[k].item(), rtol=1e-12)
    np.testing.assert_allclose(R.total_loss(obj, hp), tl.item(), rtol=1e-12)
    np.testing.assert_allclose(dnet, nt.grad.numpy(), rtol=1e-9, atol=1e-14)


def test_loss_analytic_known_answers():
    """zero logits + no objects -> iou_normal = 0.25, others 0; one GT matching
    anchor k at a cell centre -> closed forms."""
    classes, anchors = 20, np.array([[1.08, 1.19], [3.42, 4.41], [6.63, 11.38], [9.42, 5.11], [16.62, 10.52]], np.float32)
    net = np.zeros((2, 13, 13, 5 * 25), np.float32)
    empty = R.transform_labels(np.zeros(0, int), np.zeros((0, 4), np.float32), classes, 13, 13)
    labels = tuple(np.stack([e, e]) for e in empty)
    m = R.model_decode(net, classes, anchors, training=True)
    obj, _ = R.objectives(m, labels)
    assert obj['iou_best'] == 0 and obj['coords'] == 0 and obj['prob'] == 0
    np.testing.assert_allclose(obj['iou_normal'], 0.25, rtol=1e-6)
    # one object whose box equals anchor 1 centred in cell (6,6)
    k = 1
    w, h = anchors[k] / 13
    cx = cy = (6 + 0.5) / 13
    one = R.transform_labels(np.array([3]), np.array([[cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2]], np.float32), classes, 13, 13)
    labels = tuple(np.stack([o, e]) for o, e in zip(one, empty))
    obj, aux = R.objectives(m, labels)
    cnt = 2 * 169 * 5
    assert aux['mask_best'].sum() == 1 and aux['mask_best'][0, 6 * 13 + 6, k] == 1
    np.testing.assert_allclose(obj['iou_best'], 0.25 / cnt, rtol=1e-5)
    np.testing.assert_allclose(obj['iou_normal'], 0.25 * (cnt - 1) / cnt, rtol=1e-5)
    np.testing.assert_allclose(obj['coords'], 0, atol=1e-9)          # sigmoid(0)=.5 offset, wh = anchor
    np.testing.assert_allclose(obj['prob'], ((1 - 1 / 20) ** 2 + 19 / 400) / cnt, rtol=1e-5)


@pytest.mark.parametrize('name', ['darknet', 'tiny'])
def test_network_backward_matches_torch_autograd(name):
    """Whole-network gradient (conv/BN/leaky/pool/reorg/concat + loss) vs torch fp64 autograd."""
    rng = np.random.RandomState(3)
    classes, anchors = 3, np.array([[1.0, 1.2], [2.5, 3.0]])
    hp = {'prob': 1., 'iou_best': 5., 'iou_normal': 1., 'coords': 1.}
    spec = R.SPECS[name](classes, len(anchors))
    # shrink channel widths so the fp64 run is fast but the topology is intact
    spec = [(o[0], o[1], o[2], max(2, o[3] // 16) if o[4] else o[3], o[4]) if o[0] == 'conv' else o for o in spec]
    params = R.init_params(spec, seed=1, dtype=np.float64, tiny=(name == 'tiny'))
    for k in params:
        if k.endswith('gamma'):
            params[k] = rng.rand(*params[k].shape) + 0.5
        if k.endswith(('beta', 'biases')):
            params[k] = rng.randn(*params[k].shape) * 0.1
    x = rng.randn(2, 64, 64, 3)
    labels = make_labels(rng, 2, classes, 2, 2)
    net, caches = R.network_forward(spec, params, x, training=True)
    assert net.shape == (2, 2, 2, len(anchors) * (5 + classes))
    m = R.model_decode(net, classes, anchors, training=True)
    obj, aux = R.objectives(m, labels)
    grads = R.network_backward(spec, params, caches, R.loss_backward(m, labels, aux, hp, classes))

    tp = {k: _t(v).requires_grad_(True) for k, v in params.items() if k in R.trainable_names(params)}
    t = _t(x).permute(0, 3, 1, 2)
    mark = None
    for op in spec:
        if op[0] == 'conv':
            _, nm, k, cout, bn = op
            t = F.conv2d(t, tp[nm + '/weights'].permute(3, 2, 0, 1), padding=k // 2)
            if bn:
                t = F.batch_norm(t, None, None, tp[nm + '/BatchNorm/gamma'], tp[nm + '/BatchNorm/beta'], training=True, eps=1e-5)
                t = torch.maximum(t, 0.1 * t)
            else:
                t = t + tp[nm + '/biases'].reshape(1, -1, 1, 1)
        elif op[0] == 'pool':
            t = F.max_pool2d(t, 2) if op[1] == 2 else F.max_pool2d(F.pad(t, (0, 1, 0, 1), value=-np.inf), 2, stride=1)
        elif op[0] == 'mark':
            mark = t
        elif op[0] == 'reorg_concat':
            b, c, h, w = mark.shape
            r = mark.permute(0, 2, 3, 1).reshape(b, h // 2, 2, w // 2, 2, c).permute(0, 1, 3, 2, 4, 5).reshape(b, h // 2, w // 2, 4 * c)
            t = torch.cat([r.permute(0, 3, 1, 2), t], 1)
    tl, _ = _torch_loss(t.permute(0, 2, 3, 1), labels, classes, anchors, hp)
    tl.backward()
    np.testing.assert_allclose(net, t.detach().permute(0, 2, 3, 1).numpy(), rtol=1e-8, atol=1e-10)
    for k in tp:
        np.testing.assert_allclose(grads[k], tp[k].grad.numpy(), rtol=1e-6, atol=1e-12, err_msg=k)


def test_full_spec_shapes_and_macs():
    spec = R.darknet_spec(20, 5)
    convs = [o for o in spec if o[0] == 'conv']
    assert len(convs) == 22 and convs[-1][3] == 125 and convs[-2][1] == 'conv20'
    params = R.init_params(spec)
    assert params['conv20/weights'].shape == (3, 3, 3072, 1024)
    n = sum(v.size for v in params.values())
    assert n == 67_160_000 + (n - 67_160_000) and abs(n - 67.16e6) < 0.05e6   # 67.16 M incl. BN
    tiny = R.init_params(R.tiny_spec(20, 5), tiny=True)
    assert abs(sum(v.size for v in tiny.values()) - 15.87e6) < 0.05e6


def test_adam_matches_torch_formula_except_epsilon_placement():
    rng = np.random.RandomState(0)
    w, g = rng.randn(50), rng.randn(50)
    m, v = np.zeros(50), np.zeros(50)
    w1, m1, v1 = R.adam_step(w, g, m, v, 1e-3, 1)
    # t=1: m=(1-b1)g, v=(1-b2)g^2, alpha=lr*sqrt(1-b2)/(1-b1) -> step = lr*g/(|g|+eps*sqrt(1-b2)...)
    alpha = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    np.testing.assert_allclose(w1, w - alpha * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8), rtol=1e-12)
    assert R.exponential_decay(1.0, 250000, 100000, 0.96, True) == 0.96 ** 2


def test_torch_cpu_baseline_leg_matches_numpy_oracle():
    """oracle/torch_cpu_ref.py (bench.py's torch-CPU conv-stack baseline leg) evaluates the same network as the NumPy
    restatement: same logits on the same weights (training-mode BN), both model families' ops incl. the stride-1 pool."""
    from oracle import torch_cpu_ref as T
    for name in ('darknet', 'tiny'):
        spec = R.SPECS[name](3, 2)
        params = R.init_params(spec, seed=5, tiny=name == 'tiny')
        rng = np.random.RandomState(0)
        x = rng.randn(2, 64, 64, 3).astype(np.float32)
        net, _ = R.network_forward(spec, params, x, training=True)
        got = T.forward(T.build(spec, params), torch.from_numpy(x)).detach().numpy()
        assert np.abs(got - net).max() <= 2e-4 * np.abs(net).max(), name


def test_ftrl_known_answer_and_l1_zeroing():
    """TF-1.0 ApplyFtrl on a hand-computed scalar (see the arithmetic in the comments) and the |linear| <= l1 -> 0 rule."""
    w, acc, lin = np.float64([1.0]), np.float64([0.1]), np.float64([0.0])
    g = np.float64([2.0])
    # new_accum = 4.1; linear = 2 - (sqrt(4.1) - sqrt(0.1)) / 0.5 * 1; x = 0.5 * sign(linear) - linear; y = sqrt(4.1) / 0.5 + 2 * 0.25
    lin_ref = 2.0 - (np.sqrt(4.1) - np.sqrt(0.1)) / 0.5
    w_ref = (0.5 * np.sign(lin_ref) - lin_ref) / (np.sqrt(4.1) / 0.5 + 0.5)
    w1, a1, l1 = R.ftrl_step(w, g, acc, lin, 0.5, -0.5, 0.5, 0.25)
    assert abs(w1[0] - w_ref) < 1e-12 and abs(a1[0] - 4.1) < 1e-12 and abs(l1[0] - lin_ref) < 1e-12
    assert abs(w_ref - 0.2016042) < 1e-6
    w2, _, _ = R.ftrl_step(w, g, acc, lin, 0.5, -0.5, 5.0, 0.0)       # |linear| = 1.417 <= l1 = 5 -> exactly zero
    assert w2[0] == 0.0
    # general power agrees with the sqrt form at p = -0.5 (two code paths of the kernel)
    rng = np.random.RandomState(0)
    w, g = rng.randn(100), rng.randn(100)
    a = R.ftrl_step(w, g, np.full(100, 0.1), np.zeros(100), 0.1, -0.5, 0.01, 0.02)
    b = R.ftrl_step(w, g, np.full(100, 0.1), np.zeros(100), 0.1, -0.5000000001, 0.01, 0.02)
    assert np.allclose(a[0], b[0], rtol=1e-7, atol=1e-9)

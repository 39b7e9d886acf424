"""YOLO (v1) family (reference model/yolo/__init__.py:37-123, model/yolo/inference.py:24-66) through the engine and the
C ABI against the oracle: forward / loss / backward / regulariser of the `tiny` plugin with fixed dropout masks, the detection block
+ NMS, and the pieces only this family uses (dropout statistics, leaky backward)."""
import ctypes
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle import yolo2_ref as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HP = {'prob': 1., 'iou_best': 1., 'iou_normal': .5, 'coords': 5.}


def make_builder(names, size, training, basedir):
    from yolo_tf_amd import utils
    from yolo_tf_amd.model import yolo
    cfg = utils.make_config([os.path.join(ROOT, 'config.ini'), os.path.join(ROOT, 'config', 'yolo', 'tiny-%d.ini' % names)], basedir)
    cfg.set('cache', 'names', os.path.join(ROOT, cfg.get('cache', 'names')))
    cfg.set('yolo', 'width', str(size))
    cfg.set('yolo', 'height', str(size))
    utils.ensure_names(cfg)
    b = yolo.Builder(None, cfg)
    b(None, training=training)
    if training:
        b.create_objectives()
    return b


def rel(got, ref):
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))


@pytest.fixture(scope='module')
def basedir():
    with tempfile.TemporaryDirectory() as d:
        yield d


def strip(params, scope='yolo_tiny'):
    return {k[len(scope) + 1:]: v for k, v in params.items()}


@pytest.mark.parametrize('dtype,B', [('f32', 2), ('bf16', 8)])
def test_yolo1_train_step_matches_oracle(basedir, dtype, B):
    from yolo_tf_amd.session import TrainSession
    from yolo_tf_amd.utils import data
    classes, size, boxes = 20, 192, 2          # 192 / 64 = 3x3 cells: the fully connected head scales with the grid
    b = make_builder(classes, size, True, basedir)
    assert b.hparam == HP
    sess = TrainSession(b, B, dtype=dtype, optimizer='adam', learning_rate=1e-3, seed=2)
    e = sess.engine
    params0 = strip(e.get_variables())
    rng = np.random.RandomState(0)
    for k in list(params0):
        if k.endswith('biases'):
            params0[k] = (rng.randn(*params0[k].shape) * 0.1).astype(np.float32)
    e.set_variables({'yolo_tiny/' + k: v for k, v in params0.items()})
    cells = size // 64
    spec = R.yolo1_tiny_spec(classes, boxes, cells * cells)
    masks = {'dropout0': (rng.rand(B, 256) < 0.5).astype(np.uint8), 'dropout1': (rng.rand(B, 4096) < 0.5).astype(np.uint8)}
    e.dropout_masks = {'yolo_tiny/' + k: torch.from_numpy(v.reshape(-1)).cuda() for k, v in masks.items()}
    images = rng.uniform(0, 255, (B, size, size, 3)).astype(np.float32)
    labels = data.synthetic_batch(B, classes, cells, cells, seed=5)
    sess.upload_labels(labels)
    sess.forward_backward(torch.from_numpy(images).cuda())
    got = sess.fetch()
    out = e.output()
    width = out.c
    net = e.act[out][0].float().cpu().numpy().reshape(B, -1)[:, :width]
    grads = strip(e.get_gradients())

    f32 = dtype == 'f32'
    q = None if f32 else R.bf16_round
    x = np.stack([R.per_image_standardization(i) for i in images]).astype(np.float32)
    net_ref, caches = R.yolo1_forward(spec, params0, x, masks, quant=q)
    m = R.yolo1_model_decode(net_ref, classes, boxes, cells, cells, training=True)
    obj, aux = R.yolo1_objectives(m, labels)
    loss = float(sum(float(obj[k]) * HP[k] for k in R.OBJECTIVE_KEYS))
    dnet = R.yolo1_loss_backward(m, labels, aux, HP, classes, boxes, width)
    gref, reg = R.yolo1_backward(spec, params0, caches, dnet, quant=q)
    tol = 1e-4 if f32 else 3e-2
    assert rel(net, net_ref) <= tol, rel(net, net_ref)
    for k in R.OBJECTIVE_KEYS:
        assert abs(got[k] - float(obj[k])) <= (1e-4 if f32 else 5e-2) * abs(float(obj[k])) + 1e-7, (k, got[k], obj[k])
    assert abs(got['regularization'] - reg) <= 1e-5 * reg
    assert abs(got['total_loss'] - (loss + reg)) <= (1e-4 if f32 else 2e-2) * (loss + reg)
    worst = max((float(np.linalg.norm((grads[k] - gref[k]).astype(np.float64)) / (np.linalg.norm(gref[k].astype(np.float64)) + 1e-30)), k) for k in gref)
    print('yolo1 %s: net rel %.2e, worst gradient rel-L2 %s' % (dtype, rel(net, net_ref), worst))
    assert worst[0] <= (1e-4 if f32 else 0.2), worst            # measured: f32 1.3e-6, bf16 0.10 (conv0: bf16 image + eight bf16 layers of cotangent)
    if f32:
        cs = min(float((grads[k].astype(np.float64).ravel() @ gref[k].astype(np.float64).ravel()) /
                       (np.linalg.norm(grads[k].astype(np.float64)) * np.linalg.norm(gref[k].astype(np.float64)) + 1e-300)) for k in gref)
        assert cs >= 0.9995, cs
    sess.apply_gradients()
    assert sess.global_step == 1 and torch.isfinite(e.params).all()


def test_yolo1_detect_matches_oracle(basedir):
    from yolo_tf_amd.session import DetectSession
    classes, size, boxes, B = 20, 192, 2, 2
    b = make_builder(classes, size, False, basedir)
    sess = DetectSession(b, B, dtype='f32', seed=3)
    params = strip(sess.engine.get_variables())
    rng = np.random.RandomState(1)
    for k in list(params):
        if k.endswith('biases'):
            params[k] = (rng.randn(*params[k].shape) * 0.1).astype(np.float32)
    sess.engine.set_variables({'yolo_tiny/' + k: v for k, v in params.items()})
    images = rng.uniform(0, 255, (B, size, size, 3)).astype(np.float32)
    conf, mn, mx = [t.clone() for t in sess.run(torch.from_numpy(images).cuda())]
    cells = size // 64
    x = np.stack([R.per_image_standardization(i) for i in images]).astype(np.float32)
    net, _ = R.yolo1_forward(R.yolo1_tiny_spec(classes, boxes, cells * cells), params, x, None)
    m = R.yolo1_model_decode(net, classes, boxes, cells, cells, training=False)
    n = cells * cells
    assert rel(conf.cpu().numpy().reshape(B, n, boxes, classes), m['conf']) <= 1e-4
    assert rel(mn.cpu().numpy().reshape(B, n, boxes, 2), m['xy_min']) <= 1e-4
    assert rel(mx.cpu().numpy().reshape(B, n, boxes, 2), m['xy_max']) <= 1e-4
    assert rel(b.model.conf.cpu().numpy(), m['conf']) <= 1e-4 and b.model.xy_min.shape == (B, n, boxes, 2)
    # on-GPU NMS on these scores == the C oracle, bit for bit (the reference's detect.py path is model-family agnostic)
    thr = float(np.percentile(conf.cpu().numpy(), 80))
    order = sess.nms(thr, 0.4).cpu().numpy()
    lib = ctypes.CDLL(os.path.join(ROOT, 'oracle', 'libnms_ref.so'))
    P = ctypes.POINTER(ctypes.c_float)
    for i in range(B):
        c = conf[i].cpu().numpy().copy()
        o = np.zeros(c.shape[0], np.int64)
        lib.nms_ref(c.ctypes.data_as(P), mn[i].cpu().numpy().ctypes.data_as(P), mx[i].cpu().numpy().ctypes.data_as(P), ctypes.c_long(c.shape[0]),
                    ctypes.c_long(classes), ctypes.c_float(thr), ctypes.c_float(0.4), o.ctypes.data_as(ctypes.POINTER(ctypes.c_long)))
        assert np.array_equal(sess.conf[i].cpu().numpy(), c) and np.array_equal(order[i], o)


def test_yolo1_loss_kernel_and_small_kernels():
    from yolo_tf_amd import ops
    ops._lib.load()
    rng = np.random.RandomState(3)
    B, ch, cw, boxes, C = 3, 3, 4, 2, 5
    cells = ch * cw
    width = cells * (C + boxes * 5)
    ld = ops.pad8(width)
    net = np.zeros((B, ld), np.float32)
    net[:, :width] = rng.randn(B, width).astype(np.float32) * 0.5
    from yolo_tf_amd.utils import data
    labels = data.synthetic_batch(B, C, cw, ch, seed=4)
    m = R.yolo1_model_decode(net[:, :width], C, boxes, ch, cw, training=True)
    obj, aux = R.yolo1_objectives(m, labels)
    dref = R.yolo1_loss_backward(m, labels, aux, HP, C, boxes, width)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    objd = torch.zeros(4, device='cuda')
    dnet = torch.full((B * ld,), 7.0, device='cuda')
    ws = torch.zeros(ops.loss_ws_floats(B, cells, boxes), device='cuda')
    ops.yolo1_loss(dev(net).reshape(-1), ld, [dev(l) for l in labels], [HP[k] for k in R.OBJECTIVE_KEYS], objd, dnet, ws, B, ch, cw, boxes, C)
    torch.cuda.synchronize()
    for i, k in enumerate(R.OBJECTIVE_KEYS):
        assert abs(float(objd[i]) - float(obj[k])) <= 1e-5 * abs(float(obj[k])) + 1e-9, k
    d = dnet.cpu().numpy().reshape(B, ld)
    assert np.all(d[:, width:] == 0) and rel(d[:, :width], dref) <= 1e-5
    # dropout: keep rate, scaling, determinism per seed, backward = same mask; leaky backward from the output sign
    n = 1 << 20
    x = torch.randn(n, device='cuda')
    y, mask = torch.zeros(n, device='cuda'), torch.zeros(n, dtype=torch.uint8, device='cuda')
    ops.dropout(x, y, mask, n, 0.5, 12345)
    keep = float(mask.float().mean())
    assert abs(keep - 0.5) < 4 * 0.5 / np.sqrt(n) and torch.equal(y, torch.where(mask.bool(), x * 2, torch.zeros_like(x)))
    y2, mask2 = torch.zeros(n, device='cuda'), torch.zeros(n, dtype=torch.uint8, device='cuda')
    ops.dropout(x, y2, mask2, n, 0.5, 12345)
    assert torch.equal(mask, mask2)
    ops.dropout(x, y2, mask2, n, 0.5, 12346)
    assert 0.45 < float((mask != mask2).float().mean()) < 0.55
    g, dx = torch.randn(n, device='cuda'), torch.zeros(n, device='cuda')
    ops.dropout_bwd(g, mask, dx, n, 0.5)
    assert torch.equal(dx, torch.where(mask.bool(), g * 2, torch.zeros_like(g)))
    a = torch.randn(n, device='cuda').to(torch.bfloat16)
    ga, dz = torch.randn(n, device='cuda').to(torch.bfloat16), torch.zeros(n, dtype=torch.bfloat16, device='cuda')
    ops.leaky_bwd(a, ga, dz, n, 0.1)
    ref = torch.where(a.float() >= 0, ga.float(), 0.1 * ga.float()).to(torch.bfloat16)
    assert torch.equal(dz, ref)
    w, gr, loss = torch.randn(100003, device='cuda'), torch.ones(100003, device='cuda'), torch.zeros(1, dtype=torch.float64, device='cuda')
    ops.l2_regularizer(w, gr, 100003, 0.001, loss)
    assert torch.allclose(gr, 1 + 0.001 * w) and abs(float(loss) - 0.0005 * float((w.double() ** 2).sum())) < 1e-6 * float(loss)     # the scale is an f32 argument

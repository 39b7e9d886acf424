"""End-to-end parity on the MI355X: the traced Darknet-19 / Tiny YOLOv2 graphs through the engine
(forward, loss, backward, Adam, BN moving averages, detect + NMS) vs the CPU oracle on identical
weights and inputs.  Small spatial size so the NumPy oracle finishes in seconds; full-size
(416x416, batch 16) runs are checked through size-independent properties."""
import ctypes
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle import yolo2_ref as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HP = {'prob': 1., 'iou_best': 5., 'iou_normal': 1., 'coords': 1.}


def make_builder(inference, names, size, training, basedir):
    from yolo_tf_amd import utils
    from yolo_tf_amd.model import yolo2
    cfg = utils.make_config([os.path.join(ROOT, 'config.ini'), os.path.join(ROOT, 'config', 'yolo2', '%s-%d.ini' % (inference.strip('_'), names))], basedir)
    cfg.set('cache', 'names', os.path.join(ROOT, cfg.get('cache', 'names')))
    cfg.set('yolo2', 'anchors', os.path.join(ROOT, cfg.get('yolo2', 'anchors')))
    cfg.set('yolo2', 'width', str(size))
    cfg.set('yolo2', 'height', str(size))
    cfg.set('yolo2', 'inference', inference)        # (`_darknet` / `_tiny`: the biases-instead-of-beta plugins share their base's overlay)
    utils.ensure_names(cfg)
    b = yolo2.Builder(None, cfg)
    b(None, training=training)
    if training:
        b.create_objectives()
    return b, cfg


def rel(got, ref):
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))


def strip(params, scope):
    return {k[len(scope) + 1:]: v for k, v in params.items()}


@pytest.fixture(scope='module')
def basedir():
    with tempfile.TemporaryDirectory() as d:
        yield d


def cosine(a, b):
    a, b = a.astype(np.float64).ravel(), b.astype(np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


def rel_l2(a, b):
    return float(np.linalg.norm((a - b).astype(np.float64)) / (np.linalg.norm(b.astype(np.float64)) + 1e-300))


# Whole-network gradients are compared in relative L2 / cosine, not max-norm: leaky ReLU has a kink at 0, the
# forward values agree to ~2e-5 (f32) / ~1e-2 (bf16) after 22 layers, so a few pre-activations with |z| below
# that flip their slope between the two implementations and change single elements of a gradient by O(1).
# That is inherent to comparing two correct implementations of a kinked network; the exact-input backward
# kernels are pinned at 1e-4 in test_kernels_gpu.py.  Sizes keep >= 50 samples per channel in the deepest
# batch statistics (fewer makes batch norm itself ill-conditioned); the 22-layer bf16 case runs at the real 416x416
# resolution (338 samples per channel in the 13x13 stages): at toy sizes bf16 rounding noise is amplified by the
# poorly conditioned batch statistics of every layer and the comparison says little about the kernels.
@pytest.mark.parametrize('inference,size,dtype,B', [('darknet', 160, 'f32', 2), ('tiny', 160, 'f32', 2), ('darknet', 224, 'f32', 1),
                                                    ('darknet', 224, 'bf16', 2), ('tiny', 160, 'bf16', 4)])
def test_train_step_matches_oracle(basedir, inference, size, dtype, B):
    from yolo_tf_amd.session import TrainSession
    from yolo_tf_amd.utils import data
    classes = 20
    b, cfg = make_builder(inference, classes, size, True, basedir)
    sess = TrainSession(b, B, dtype=dtype, optimizer='adam', learning_rate=1e-3, seed=3)
    scope = 'yolo2_' + inference
    params0 = strip(sess.engine.get_variables(), scope)
    # non-trivial BN parameters so that beta/gamma gradients are exercised
    rng = np.random.RandomState(0)
    for k in list(params0):
        if k.endswith('gamma'):
            params0[k] = (rng.rand(*params0[k].shape) + 0.5).astype(np.float32)
        if k.endswith(('beta', 'biases')):
            params0[k] = (rng.randn(*params0[k].shape) * 0.1).astype(np.float32)
    sess.engine.set_variables({scope + '/' + k: v for k, v in params0.items()})
    cells = size // 32
    images = rng.uniform(0, 255, (B, size, size, 3)).astype(np.float32)
    labels = data.synthetic_batch(B, classes, cells, cells, seed=7)
    sess.step(torch.from_numpy(images).cuda(), labels)
    got = sess.fetch()
    e = sess.engine
    logits = e.act[e.output()][0].float().cpu().numpy().reshape(B, cells, cells, -1)[..., :b.model.inputs.c]
    grads = strip(e.get_gradients(), scope)
    params1 = strip(e.get_variables(), scope)

    spec = R.SPECS[inference](classes, len(b.anchors))
    x = np.stack([R.per_image_standardization(i) for i in images]).astype(np.float32)
    f32 = dtype == 'f32'
    # bf16 mode = f32 arithmetic + bf16 storage: the oracle rounds to bf16 at exactly the product's storage points
    new_params, _, info = R.train_step(spec, params0, {}, x, labels, classes, b.anchors, HP, 1e-3, 0, quant=None if f32 else R.bf16_round)
    tol_out, tol_loss = (1e-4, 1e-4) if f32 else (0.2, 3e-2)
    r = rel(logits, info['net'])
    l2 = sorted(((rel_l2(grads[k], info['grads'][k]), k) for k in grads), reverse=True)
    cs = sorted((cosine(grads[k], info['grads'][k]), k) for k in grads)
    print('%s %d %s: logits rel %.2e, loss %.6f vs %.6f; worst grad rel-L2 %s; worst cosine %s'
          % (inference, size, dtype, r, got['total_loss'], info['loss'], ['%s %.2e' % (k, v) for v, k in l2[:3]], ['%s %.5f' % (k, v) for v, k in cs[:3]]))
    assert r <= tol_out, 'logits rel err %.3e' % r
    for k in R.OBJECTIVE_KEYS:
        assert abs(got[k] - info['objectives'][k]) <= (tol_loss if f32 else 0.3) * abs(info['objectives'][k]) + 1e-7, (k, got[k], info['objectives'][k])
    assert abs(got['total_loss'] - info['loss']) <= tol_loss * abs(info['loss'])
    if f32:
        assert l2[0][0] <= 2e-2, 'worst gradient rel-L2 err %.3e at %s' % l2[0]
        assert cs[0][0] >= 0.9995, 'worst gradient cosine %.5f at %s' % cs[0]
    else:
        # bf16 end to end is NOT an equality test.  Batch-norm backward outputs sum to zero per channel, so the filter
        # gradient sum_m x[m]*dy[m] is a near-cancelling sum; at random init its signal (weak activation/gradient
        # correlation) is comparable to the residue mu_x * sum(rounding errors of dy), and two valid bf16 computations
        # (this engine, the bf16-storage oracle) diverge chaotically layer by layer (scripts/debug_bwd.py shows 1-ulp
        # agreement in the first layers growing ~1.3x per layer).  Equality is pinned per kernel (test_kernels_gpu.py,
        # bf16 cases) and for the whole network in f32 (above); here: same loss, same gradient scale, same direction.
        ratio = sorted((np.linalg.norm(grads[k].astype(np.float64)) / (np.linalg.norm(info['grads'][k].astype(np.float64)) + 1e-300), k) for k in grads)
        med = float(np.median([c for c, _ in cs]))
        print('bf16: median gradient cosine %.3f, min %.3f (%s); gradient norm ratio in [%.2f, %.2f]' % (med, cs[0][0], cs[0][1], ratio[0][0], ratio[-1][0]))
        assert med >= 0.5 and cs[0][0] >= 0.2, (med, cs[0])
        assert 0.5 <= ratio[0][0] and ratio[-1][0] <= 2.0, (ratio[0], ratio[-1])
    if f32:
        # Adam moves every weight by ~lr at step 1 regardless of gradient magnitude, so compare the update direction
        for k in ('conv0/weights', 'conv/weights', 'conv/biases'):
            du_g, du_r = params1[k] - params0[k], new_params[k] - params0[k]
            agree = np.mean(np.sign(du_g) == np.sign(du_r))
            assert agree > 0.98, (k, agree)
        for k in params0:
            if k.endswith(('moving_mean', 'moving_variance')):
                assert rel(params1[k], new_params[k]) <= 1e-4, k


@pytest.mark.parametrize('inference,classes,B,size', [('darknet', 20, 2, 96), ('tiny', 20, 2, 96), ('darknet', 80, 2, 96), ('tiny', 80, 2, 96),
                                                      ('tiny', 20, 1, 416)],      # BASELINE configs[0] at its stated size: ONE 416 x 416 image
                         ids=['darknet-20', 'tiny-20', 'darknet-80', 'tiny-80', 'tiny-20-416x416-batch1'])
def test_detect_matches_oracle(basedir, inference, classes, B, size):
    """Inference mode (BASELINE configs[0] is Tiny-YOLOv2 VOC-20 detect, one 416 x 416 image: the last case; reference detect.py:59-91):
    moving-average BN folded into the filters, the tiny model's stride-1 SAME pool, decode, NMS -- against the oracle's unfolded network on
    the same weights, then the C restatement of utils/postprocess.py:39-51 on the very same scores."""
    from yolo_tf_amd.session import DetectSession
    b, _ = make_builder(inference, classes, size, False, basedir)
    sess = DetectSession(b, B, dtype='f32', seed=5)
    scope = 'yolo2_' + inference
    params = strip(sess.engine.get_variables(), scope)
    rng = np.random.RandomState(1)
    for k in list(params):          # moving stats away from their init so inference-mode BN is exercised
        if k.endswith('moving_mean'):
            params[k] = (rng.randn(*params[k].shape) * 0.05).astype(np.float32)
        if k.endswith('moving_variance'):
            params[k] = (rng.rand(*params[k].shape) + 0.5).astype(np.float32)
    params['conv/biases'] = (rng.randn(*params['conv/biases'].shape)).astype(np.float32)
    if inference == 'tiny':
        # the tiny plugin's truncated_normal(0.1) filters (reference model/yolo2/inference.py:33) amplify ~3000x through nine
        # un-normalised layers in inference mode: exp() of the box logits overflows and check_numerics raises (correctly, and
        # so would the reference).  Trained weights do not do that: use fan-in-scaled filters.
        for k in list(params):
            if k.endswith('/weights'):
                kh, kw, cin, _ = params[k].shape
                params[k] = (rng.randn(*params[k].shape) * np.sqrt(1.0 / (kh * kw * cin))).astype(np.float32)
    sess.engine.set_variables({scope + '/' + k: v for k, v in params.items()})
    images = rng.uniform(0, 255, (B, size, size, 3)).astype(np.float32)
    conf, mn, mx = [t.clone() for t in sess.run(torch.from_numpy(images).cuda())]
    x = np.stack([R.per_image_standardization(i) for i in images]).astype(np.float32)
    net, _ = R.network_forward(R.SPECS[inference](classes, 5), params, x, training=False)
    m = R.model_decode(net, classes, b.anchors, training=False)
    cells = (size // 32) ** 2
    assert rel(conf.cpu().numpy().reshape(B, cells, 5, classes), m['conf']) <= 1e-4
    assert rel(mn.cpu().numpy().reshape(B, cells, 5, 2), m['xy_min']) <= 1e-4
    assert rel(mx.cpu().numpy().reshape(B, cells, 5, 2), m['xy_max']) <= 1e-4
    # the attributes reference callers read off the model object (detect.py:69-72, demo_detect.py:62)
    mdl = b.model
    for key in ('conf', 'xy_min', 'xy_max', 'iou', 'prob', 'xy', 'wh'):
        got = getattr(mdl, key).cpu().numpy()
        assert got.shape == m[key].shape, (key, got.shape, m[key].shape)
        assert rel(got, m[key]) <= 1e-4, key
    # NMS on the GPU-produced scores must equal the C oracle on the very same scores, bit for bit
    thr = float(np.percentile(conf.cpu().numpy(), 90))
    order = sess.nms(thr, 0.4).cpu().numpy()
    lib = ctypes.CDLL(os.path.join(ROOT, 'oracle', 'libnms_ref.so'))
    P = ctypes.POINTER(ctypes.c_float)
    for i in range(B):
        c = conf[i].cpu().numpy().copy()
        o = np.zeros(c.shape[0], np.int64)
        lib.nms_ref(c.ctypes.data_as(P), mn[i].cpu().numpy().ctypes.data_as(P), mx[i].cpu().numpy().ctypes.data_as(P), ctypes.c_long(c.shape[0]),
                    ctypes.c_long(classes), ctypes.c_float(thr), ctypes.c_float(0.4), o.ctypes.data_as(ctypes.POINTER(ctypes.c_long)))
        assert np.array_equal(sess.conf[i].cpu().numpy(), c)
        assert np.array_equal(order[i], o)


def test_detect_raises_on_non_finite_outputs_like_check_numerics(basedir):
    """detect.py:70 wraps the model outputs in tf.check_numerics: NaN/Inf raises.  The tiny plugin at its own initialisation
    (truncated_normal(0.1) filters, identity moving statistics) overflows exp() of the box logits -- a natural way to force it."""
    from yolo_tf_amd.session import DetectSession
    b, _ = make_builder('tiny', 20, 96, False, basedir)
    sess = DetectSession(b, 1, dtype='f32', seed=5)
    images = torch.rand(1, 96, 96, 3, device='cuda') * 255
    with pytest.raises(FloatingPointError, match='NaN or Inf'):
        sess.run(images)
    sess.run(images, check_numerics=False)          # the flag is advisory when the caller opts out (benchmarks)


def test_postprocess_dropin_contract(golden_dir):
    """utils.postprocess.non_max_suppress: same signature, in-place mutation, returned views and order."""
    from yolo_tf_amd.utils import postprocess
    g = np.load(os.path.join(golden_dir, 'nms.npz'))
    for case in ('sparse20', 'ties', 'identical_ties'):
        conf = g[case + '/conf_in'].copy()
        mn, mx = g[case + '/xy_min'], g[case + '/xy_max']
        boxes = postprocess.non_max_suppress(conf, mn, mx, float(g[case + '/thr']), float(g[case + '/thr_iou']))
        assert np.array_equal(conf, g[case + '/conf_out'])                     # caller's array mutated
        n, C = conf.shape[0] * conf.shape[1], conf.shape[2]
        flat = conf.reshape(n, C)
        for (c_row, b_min, b_max), idx in zip(boxes, g[case + '/order']):
            assert np.shares_memory(c_row, conf) and np.array_equal(c_row, flat[idx])
            assert np.array_equal(b_min, mn.reshape(n, 2)[idx]) and np.array_equal(b_max, mx.reshape(n, 2)[idx])


def test_full_size_training_properties(basedir):
    """BASELINE config #2 shape (Darknet-19, VOC-20, 416x416, batch 16, bf16): finite loss that goes
    down over a few Adam steps on a fixed batch, BN moving averages move, padding lanes stay zero."""
    from yolo_tf_amd.session import TrainSession
    from yolo_tf_amd.utils import data
    B = 16
    b, _ = make_builder('darknet', 20, 416, True, basedir)
    sess = TrainSession(b, B, dtype='bf16', optimizer='adam', learning_rate=1e-4, seed=0)
    g = torch.Generator(device='cuda').manual_seed(1234)
    images = torch.rand(B, 416, 416, 3, device='cuda', generator=g) * 255
    sess.upload_labels(data.synthetic_batch(B, 20, 13, 13, seed=4321))
    losses = []
    for _ in range(6):
        sess.step(images)
        losses.append(sess.fetch()['total_loss'])
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses
    e = sess.engine
    out = e.act[e.output()][0].float().reshape(B, 13, 13, 128)
    assert torch.all(out[..., 125:] == 0)
    assert float(e.var['yolo2_darknet/conv0/BatchNorm/moving_variance'].sub(1).abs().max()) > 0
    assert torch.isfinite(e.params).all() and torch.isfinite(e.grads).all()


def test_layerwise_adam_equals_the_single_launch_bit_for_bit(basedir, monkeypatch):
    """With YOLO2_EARLY_ADAM=1 TrainSession.step updates each layer's filter as soon as its gradient is final (engine.adam_update_layer on a
    third stream, then adam_update_small; off by default: measured slower, profiles/r05_early_adam.txt): on the SAME gradients the parameters, both Adam slots and both MFMA operand layouts of every layer must equal the
    one-launch form (adam_update_and_prepare) bit for bit -- and a session stepping that way must train (finite, decreasing loss)."""
    from yolo_tf_amd.session import TrainSession
    from yolo_tf_amd.utils import data
    B = 4
    monkeypatch.setenv('YOLO2_EARLY_ADAM', '1')
    b, _ = make_builder('darknet', 20, 160, True, basedir)
    sess = TrainSession(b, B, dtype='bf16', optimizer='adam', learning_rate=1e-3, seed=2)
    assert sess.early_adam
    e, opt = sess.engine, sess.optimizer
    g = torch.Generator(device='cuda').manual_seed(7)
    images = torch.rand(B, 160, 160, 3, device='cuda', generator=g) * 255
    sess.upload_labels(data.synthetic_batch(B, 20, 5, 5, seed=8))
    sess.forward_backward(images)                      # gradients only: no update (defer_collectives = False)
    torch.cuda.synchronize()
    p0, m0, v0 = e.params.clone(), opt.slots[0].clone(), opt.slots[1].clone()
    convs = [op for op in e.graph.ops if op['kind'] == 'conv']
    operands = lambda: [t.clone() for op in convs for t in (e.conv[op['name']]['Ffwd'], e.conv[op['name']].get('Fdgr')) if t is not None]
    args = (3e-4, 0.9, 0.999, 1e-8, 1.0)
    e.adam_update_and_prepare(opt.slots[0], opt.slots[1], *args)
    torch.cuda.synchronize()
    ref = (e.params.clone(), opt.slots[0].clone(), opt.slots[1].clone(), operands())
    e.params.copy_(p0); opt.slots[0].copy_(m0); opt.slots[1].copy_(v0)
    for op in reversed(convs):                         # backward order, as the step issues them
        e.adam_update_layer(op, opt.slots[0], opt.slots[1], *args)
    e.adam_update_small(opt.slots[0], opt.slots[1], *args)
    torch.cuda.synchronize()
    got = (e.params, opt.slots[0], opt.slots[1], operands())
    for a, r in zip(got[:3], ref[:3]):
        assert torch.equal(a, r)
    assert not torch.equal(e.params, p0)
    for a, r in zip(got[3], ref[3]):
        assert torch.equal(a, r)
    losses = []
    for _ in range(5):
        sess.step(images)
        losses.append(sess.fetch()['total_loss'])
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def _nms_c_oracle(conf, mn, mx, thr, thr_iou):
    """oracle/nms_ref.c on one image; returns (conf after, order)."""
    lib = ctypes.CDLL(os.path.join(ROOT, 'oracle', 'libnms_ref.so'))
    P = ctypes.POINTER(ctypes.c_float)
    c = np.ascontiguousarray(conf, np.float32).copy()
    mn, mx = np.ascontiguousarray(mn, np.float32), np.ascontiguousarray(mx, np.float32)
    o = np.zeros(c.shape[0], np.int64)
    lib.nms_ref(c.ctypes.data_as(P), mn.ctypes.data_as(P), mx.ctypes.data_as(P), ctypes.c_long(c.shape[0]), ctypes.c_long(c.shape[1]),
                ctypes.c_float(thr), ctypes.c_float(thr_iou), o.ctypes.data_as(ctypes.POINTER(ctypes.c_long)))
    return c, o


@pytest.mark.parametrize('B', [32, 256])
def test_full_size_detect_batch_properties(basedir, B):
    """BASELINE config #5 shape (batch-256 416x416 detect + on-GPU NMS, and batch 32): NMS is idempotent, survivors per class
    are mutually below the IoU threshold; then the reference's DENSE worst case (every box a candidate: conf ~ U(0, 0.5),
    utils/postprocess.py:39-51 spends 69.5 s per image on it) written over the decoded scores of the whole batch,
    first / middle / last image bit-exact against the C oracle."""
    from yolo_tf_amd.session import DetectSession
    b, _ = make_builder('darknet', 20, 416, False, basedir)
    sess = DetectSession(b, B, dtype='bf16', seed=0)
    images = torch.rand(B, 416, 416, 3, device='cuda') * 255
    conf, mn, mx = sess.run(images)
    assert torch.isfinite(conf).all() and torch.isfinite(mn).all() and torch.isfinite(mx).all()
    thr = float(torch.quantile(conf.flatten()[:1000000], 0.98))
    before = conf.clone()
    sess.nms(thr, 0.4)
    once = sess.conf.clone()
    assert torch.all((once == before) | (once == 0))
    sess.nms(thr, 0.4)
    assert torch.equal(sess.conf, once)                       # idempotent
    c = once[0].cpu().numpy()
    mnn, mxx = mn[0].cpu().numpy(), mx[0].cpu().numpy()
    for k in range(3):
        keep = np.where(c[:, k] > thr)[0]
        for i in keep:
            for j in keep:
                if i < j:
                    assert R.iou(mnn[i], mxx[i], mnn[j], mxx[j]) < np.float32(0.4)
    # dense stress scores on the real decoded geometry
    g = torch.Generator(device='cuda').manual_seed(5)
    sess.conf.copy_(torch.rand(sess.conf.shape, device='cuda', generator=g) * 0.5)
    dense = sess.conf.clone()
    order = sess.nms(0.3, 0.4).cpu().numpy()
    after = sess.conf.cpu().numpy()
    assert (after != dense.cpu().numpy()).any()               # the NMS really suppressed something
    for i in (0, B // 2, B - 1):
        c_ref, o_ref = _nms_c_oracle(dense[i].cpu().numpy(), mn[i].cpu().numpy(), mx[i].cpu().numpy(), 0.3, 0.4)
        assert np.array_equal(after[i], c_ref), 'image %d: scores differ from the C oracle' % i
        assert np.array_equal(order[i], o_ref), 'image %d: order differs' % i


@pytest.mark.parametrize('plugin,classes', [('_darknet', 20), ('_tiny', 80)])
def test_darknet_weights_file_to_detect(basedir, tmp_path, plugin, classes):
    """SURVEY 8f-2 GPU leg (reference parse_darknet_yolo2.py:58-117): a Darknet .weights file written by an independent
    writer in Darknet's layout -> darknet_weights.load -> Engine.set_variables -> DetectSession -> decoded boxes + NMS,
    against the oracle run on the weights as the reference's converter would interpret the same bytes."""
    import struct
    from yolo_tf_amd import darknet_weights as D
    from yolo_tf_amd.session import DetectSession
    from yolo_tf_amd import utils
    from yolo_tf_amd.model import yolo2
    B, size, A = 2, 96, 5
    cfg = utils.make_config([os.path.join(ROOT, 'config.ini'), os.path.join(ROOT, 'config', 'yolo2', '%s-%d.ini' % (plugin.strip('_'), classes))], basedir)
    cfg.set('cache', 'names', os.path.join(ROOT, cfg.get('cache', 'names')))
    cfg.set('yolo2', 'anchors', os.path.join(ROOT, cfg.get('yolo2', 'anchors')))
    cfg.set('yolo2', 'width', str(size))
    cfg.set('yolo2', 'height', str(size))
    cfg.set('yolo2', 'inference', plugin)
    utils.ensure_names(cfg)
    b = yolo2.Builder(None, cfg)
    b(None, training=False)
    scope = 'yolo2_' + plugin.strip('_')
    convs = [op for op in b.graph.ops if op['kind'] == 'conv']
    rng = np.random.RandomState(4)
    path = str(tmp_path / 'net.weights')
    oracle_params = {}
    with open(path, 'wb') as f:
        f.write(struct.pack('4i', 0, 1, 0, 777))
        for op in convs:
            k, cin, cout = op['ksize'], op['cin'], op['cout']
            name = op['name'][len(scope) + 1:]
            w = (rng.randn(cout, cin, k, k) / np.sqrt(k * k * cin)).astype(np.float32)      # OIHW on disk
            if op['bn']:
                bias, gamma = (rng.randn(cout) * 0.1).astype(np.float32), (rng.rand(cout) + 0.5).astype(np.float32)
                mm, mv = (rng.randn(cout) * 0.05).astype(np.float32), (rng.rand(cout) + 0.5).astype(np.float32)
                for v in (bias, gamma, mm, mv, w):
                    f.write(v.tobytes())
                oracle_params.update({name + '/BatchNorm/beta': bias, name + '/BatchNorm/gamma': gamma, name + '/BatchNorm/moving_mean': mm,
                                      name + '/BatchNorm/moving_variance': mv, name + '/weights': w.transpose(2, 3, 1, 0)})
            else:
                bias = rng.randn(cout).astype(np.float32)
                f.write(bias.tobytes())
                f.write(w.tobytes())
                per = cout // A                                  # Darknet head order (x,y,w,h,obj,cls..) -> (obj,x,y,w,h,cls..)
                perm = np.concatenate([np.array([4, 0, 1, 2, 3] + list(range(5, per))) + a * per for a in range(A)])
                oracle_params.update({name + '/biases': bias[perm], name + '/weights': w.transpose(2, 3, 1, 0)[..., perm]})
    header, values = D.load(path, b.graph, A)
    assert header['seen'] == 777 and header['remaining'] == 0
    sess = DetectSession(b, B, dtype='f32', seed=0)
    sess.engine.set_variables(values)
    images = rng.uniform(0, 255, (B, size, size, 3)).astype(np.float32)
    conf, mn, mx = [t.clone() for t in sess.run(torch.from_numpy(images).cuda(), preprocess_mode=1)]     # Darknet models take x/255
    x = (images / np.float32(255)).astype(np.float32)
    net, _ = R.network_forward(R.SPECS[plugin.strip('_')](classes, A), oracle_params, x, training=False)
    m = R.model_decode(net, classes, b.anchors, training=False)
    cells = (size // 32) ** 2
    assert rel(conf.cpu().numpy().reshape(B, cells, A, classes), m['conf']) <= 1e-4
    assert rel(mn.cpu().numpy().reshape(B, cells, A, 2), m['xy_min']) <= 1e-4
    assert rel(mx.cpu().numpy().reshape(B, cells, A, 2), m['xy_max']) <= 1e-4
    thr = float(np.percentile(conf.cpu().numpy(), 90))
    order = sess.nms(thr, 0.4).cpu().numpy()
    for i in range(B):
        c_ref, o_ref = _nms_c_oracle(conf[i].cpu().numpy(), mn[i].cpu().numpy(), mx[i].cpu().numpy(), thr, 0.4)
        assert np.array_equal(sess.conf[i].cpu().numpy(), c_ref) and np.array_equal(order[i], o_ref)
    # and back: export the engine's variables, byte-identical file
    out = str(tmp_path / 'out.weights')
    D.save(out, b.graph, sess.engine.get_variables(), A, header=(0, 1, 0, 777))
    assert open(out, 'rb').read() == open(path, 'rb').read()


def _dp_worker(rank, world, port, outdir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    # The test runs forward twice and compares the two backward passes to 1e-5.  The fused BN statistics accumulate tile sums
    # with f32 atomics, so their last bit depends on arrival order; with 18 samples per channel (3x3 cells, batch 2) that can
    # flip a leaky-ReLU kink between the two forwards.  The standalone statistics pass is order-independent.
    os.environ['YOLO2_FUSE_BN_STATS'] = '0'
    import torch.distributed as dist
    from yolo_tf_amd.parallel import init_distributed
    from yolo_tf_amd.session import TrainSession
    from yolo_tf_amd.utils import data
    torch.cuda.set_device(0)
    init_distributed(backend='gloo')                    # both ranks share the one GPU of the test box: gloo carries the CUDA tensors
    b, _ = make_builder('tiny', 20, 96, True, os.path.join(outdir, 'base%d' % rank))
    sess = TrainSession(b, 2, dtype='f32', optimizer='adam', learning_rate=1e-3, seed=3, world_size=world, bucket_mb=8.0)
    assert len(sess.reducer.buckets) >= 3
    rng = np.random.RandomState(100 + rank)             # different data per rank, same initial weights (same seed)
    images = torch.from_numpy(rng.uniform(0, 255, (2, 96, 96, 3)).astype(np.float32)).cuda()
    sess.upload_labels(data.synthetic_batch(2, 20, 3, 3, seed=200 + rank))
    e = sess.engine
    # local gradient first (no collective), then the data-parallel step
    sess.reducer, keep = None, sess.reducer
    sess.forward_backward(images)
    local = e.grads.clone()
    sess.reducer = keep
    sess.forward_backward(images)
    summed = e.grads.clone()
    sess.apply_gradients()
    torch.cuda.synchronize()
    params1 = e.params.clone()
    sess.step(images)                                   # the production path: buckets consumed by the optimizer as they arrive
    torch.cuda.synchronize()
    params2 = e.params.clone()
    # the bf16 wire format + per-bucket timing (what `bench.py --gpus N --grad-dtype bf16` runs): the cast kernels and the collective on
    # the communication stream, two ranks; replicas must stay identical and the report has one entry per bucket
    from yolo_tf_amd.parallel import GradReducer
    from yolo_tf_amd import ops
    assert ops.get_stream_workgroups() == max(64, torch.cuda.get_device_properties(0).multi_processor_count - 32)      # CUs left to the collective
    sess.reducer = GradReducer(e.grads, list(e.param_offsets.values()), 8.0, grad_dtype='bf16', timing=True)
    sess.step(images)
    torch.cuda.synchronize()
    rep = sess.reducer.exposed_times()
    assert len(rep) == len(sess.reducer.buckets) and all(r['collective_ms'] > 0 and r['exposed_ms'] >= 0 for r in rep)
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), local=local.cpu().numpy(), summed=summed.cpu().numpy(), params=params1.cpu().numpy(),
             params2=params2.cpu().numpy(), params3=e.params.cpu().numpy(), grads3=e.grads.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_step_two_processes_one_gpu(tmp_path):
    """GradReducer end to end on device tensors: buckets launched from the backward hook on the side stream,
    sum over ranks == sum of the ranks' local gradients, replicas stay identical after the averaged Adam step."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(str(tmp_path / 'rank0.npz')), np.load(str(tmp_path / 'rank1.npz'))
    np.testing.assert_array_equal(r0['summed'], r1['summed'])
    ref = r0['local'] + r1['local']
    assert np.abs(r0['summed'] - ref).max() <= 1e-5 * np.abs(ref).max()      # f32 atomics: order-dependent rounding only
    np.testing.assert_array_equal(r0['params'], r1['params'])
    np.testing.assert_array_equal(r0['params2'], r1['params2'])                 # bucket-wise update: replicas still identical
    np.testing.assert_array_equal(r0['params3'], r1['params3'])                 # ... and with the bf16 wire format
    np.testing.assert_array_equal(r0['grads3'], r1['grads3'])
    assert np.array_equal(r0['grads3'], torch.from_numpy(r0['grads3']).to(torch.bfloat16).float().numpy())     # what arrived are bf16 values
    assert np.abs(r0['params2'] - r0['params']).max() > 0
    assert np.abs(r0['local'] - r1['local']).max() > 0                          # the ranks really saw different data


# ---------------------------------------------------------------------------------------------------------------------
# Teacher-forced, layer-by-layer backward parity.  Comparing whole-network gradients of two implementations end to end is
# ill-posed across the leaky-ReLU kink and the max-pool arg-max (see the comment above test_train_step_matches_oracle), so
# this test removes the amplification instead of loosening the tolerance: for every layer the oracle's backward formulas
# are evaluated on the GPU's OWN stored forward state and incoming gradient (x, raw conv output y, batch moments, dA), and
# compared with what the GPU kernels produced for that layer (dgamma, dbeta, dW, dX).  Forward parity is held separately
# (logits 1e-4 above; per-kernel tests).  Every kernel of the backward sweep is then pinned at per-kernel tolerance at the
# NETWORK's real shapes -- including 416x416 batch 16 in bf16, the benchmarked configuration -- and a wrong tap, a dropped
# tile or a stale stream-K slot in any one layer fails that layer's line.
# ---------------------------------------------------------------------------------------------------------------------

def read_t(e, t, grad=False):
    """[B,h,w,c] float32 copy of an engine activation (or its gradient), honouring the pixel stride and concat aliasing."""
    buf, ld = (e.gact if grad else e.act)[t]
    v = torch.as_strided(buf, (e.B * t.h * t.w, t.c), (ld, 1))
    return v.float().cpu().numpy().reshape(e.B, t.h, t.w, t.c)


def forward_teacher_forced(e, scope, params0, f32):
    """Forward, layer by layer, on the GPU's OWN stored inputs: raw convolution output vs the oracle's convolution of the stored input,
    BN + leaky (+ fused 2x2 pool) output vs the oracle's formulas on the stored y and batch moments, stand-alone pools and reorg
    bit-exact.  A forward bug that is self-consistent (and would pass a backward comparison that starts from the GPU's forward state)
    fails here.  Returns [(rel-L2, what)] sorted worst first."""
    q = (lambda a: a) if f32 else R.bf16_round
    tol_fwd, tol_fwd_l2 = (2e-5, 1e-5) if f32 else (8e-3, 2e-3)     # bf16: one output ulp where the f32 sums round differently
    fwd_report = []
    for op in e.graph.ops:
        if op['kind'] == 'conv':
            name = op['name'][len(scope) + 1:]
            xin = read_t(e, op['x'])
            conv = R.conv2d(xin, q(params0[name + '/weights']))
            if op['bn'] and e._first_fused(op):
                y = q(conv)          # the image layer's raw output is never stored (recomputed inside its fused consumers): the oracle's stands in,
                                     # and the pooled activation below checks the fused kernel against it
            elif op['bn']:
                y = read_t(e, op['y'])
                fwd_report.append((rel_l2(y, q(conv)), name + ' y'))
                assert rel(y, q(conv)) <= tol_fwd and fwd_report[-1][0] <= tol_fwd_l2, '%s conv output: max %.3e l2 %.3e' % (name, rel(y, q(conv)), fwd_report[-1][0])
            if op['bn']:
                st = e.conv[op['name']]
                bname = name + ('/BatchNorm/beta' if (name + '/BatchNorm/beta') in params0 else '/biases')
                a = q(R.leaky_relu(R.bn_apply(y, st['mean'].cpu().numpy(), st['var'].cpu().numpy(), params0[name + '/BatchNorm/gamma'], params0[bname])))
                pool = e.fused_pool.get(op['out'])
                got_a, ref_a = (read_t(e, pool['out']), R.max_pool(a, 2)) if pool is not None else (read_t(e, op['out']), a)
                fwd_report.append((rel_l2(got_a, ref_a), name + ' activation'))
                assert rel(got_a, ref_a) <= tol_fwd and fwd_report[-1][0] <= tol_fwd_l2, '%s activation: max %.3e' % (name, rel(got_a, ref_a))
                if pool is not None and st.get('ymax_valid'):
                    # the side output the backward reduction reads: the STORED raw output at the first maximum of each window, exactly
                    Bq, Hq, Wq, Cq = y.shape
                    win = lambda t: t.reshape(Bq, Hq // 2, 2, Wq // 2, 2, Cq).transpose(0, 1, 3, 2, 4, 5).reshape(Bq, Hq // 2, Wq // 2, 4, Cq)
                    got_act = q(R.leaky_relu(R.bn_apply(y, st['mean'].cpu().numpy(), st['var'].cpu().numpy(), params0[name + '/BatchNorm/gamma'], params0[bname])))
                    arg = np.argmax(win(got_act), axis=3)[:, :, :, None, :]
                    want = np.take_along_axis(win(y), arg, axis=3)[:, :, :, 0, :]
                    n = want.size
                    got_ym = st['pool_ymax'][:n].float().cpu().numpy().reshape(want.shape)
                    same = float(np.mean(got_ym == want))
                    assert same >= 0.9999, '%s ymax: %.6f equal' % (name, same)     # (a tie broken by a last-ulp difference of the activation picks the other, equal-activation, position)
            else:
                ref_o = q(conv + params0[name + '/biases'])
                assert rel(read_t(e, op['out']), ref_o) <= tol_fwd, name
        elif op['kind'] == 'pool' and op['x'] not in e.fused_pool:
            assert np.array_equal(read_t(e, op['out']), R.max_pool(read_t(e, op['x']), op['stride'])), op['name']
        elif op['kind'] == 'reorg':
            assert np.array_equal(read_t(e, op['out']), R.reorg(read_t(e, op['x']))), op['name']
    fwd_report.sort(reverse=True)
    return fwd_report



@pytest.mark.parametrize('inference,size,dtype,B,classes', [
    ('darknet', 160, 'f32', 2, 20), ('tiny', 160, 'f32', 2, 20), ('darknet', 160, 'f32', 2, 80),
    ('darknet', 416, 'bf16', 16, 20),       # BASELINE configs[1]: the benchmarked shape and dtype
    ('darknet', 416, 'bf16', 8, 80),        # per-GPU shape of BASELINE configs[2] (COCO-80, 425-wide head)
    ('tiny', 416, 'bf16', 4, 20)])
def test_backward_layerwise_teacher_forced(basedir, inference, size, dtype, B, classes):
    from yolo_tf_amd.session import TrainSession
    from yolo_tf_amd.utils import data
    lr = 1e-3
    b, cfg = make_builder(inference, classes, size, True, basedir)
    sess = TrainSession(b, B, dtype=dtype, optimizer='adam', learning_rate=lr, seed=11)
    e = sess.engine
    scope = 'yolo2_' + inference
    params0 = strip(e.get_variables(), scope)
    rng = np.random.RandomState(2)
    for k in list(params0):
        if k.endswith('gamma'):
            params0[k] = (rng.rand(*params0[k].shape) + 0.5).astype(np.float32)
        if k.endswith(('beta', 'biases')):
            params0[k] = (rng.randn(*params0[k].shape) * 0.1).astype(np.float32)
    e.set_variables({scope + '/' + k: v for k, v in params0.items()})
    cells = size // 32
    g = torch.Generator(device='cuda').manual_seed(99)
    images = torch.rand(B, size, size, 3, device='cuda', generator=g) * 255
    labels = data.synthetic_batch(B, classes, cells, cells, seed=17)
    sess.upload_labels(labels)
    sess.forward_backward(images)
    got = sess.fetch()
    torch.cuda.synchronize()
    f32 = dtype == 'f32'
    q = (lambda a: a) if f32 else R.bf16_round
    tol_vec, tol_l2 = (1e-5, 1e-5) if f32 else (5e-4, 5e-4)       # measured worst: 1e-6 (f32), 1e-4 (bf16, conv20 dX at 416x416 batch 16)
    grads = strip(e.get_gradients(), scope)

    # ---- loss and its gradient on the GPU's own logits
    out_t = e.output()
    logits = read_t(e, out_t)
    m = R.model_decode(logits, classes, b.anchors, training=True)
    obj, aux = R.objectives(m, labels)
    for k in R.OBJECTIVE_KEYS:
        assert abs(got[k] - float(obj[k])) <= 1e-4 * abs(float(obj[k])) + 1e-9, (k, got[k], obj[k])
    dnet_ref = q(R.loss_backward(m, labels, aux, HP, classes))
    dnet = read_t(e, out_t, grad=True)
    assert rel(dnet, dnet_ref) <= (1e-4 if f32 else 8e-3), 'dlogits rel err %.3e' % rel(dnet, dnet_ref)

    fwd_report = forward_teacher_forced(e, scope, params0, f32)
    print('\n%s %d %s B%d C%d teacher-forced forward, worst rel-L2: %s' % (inference, size, dtype, B, classes, ['%s %.1e' % (n, r) for r, n in fwd_report[:4]]))

    # ---- every convolution layer, in graph order
    inputs = set(e.graph.inputs.values())
    report = []
    for op in e.graph.ops:
        if op['kind'] != 'conv':
            continue
        name = op['name'][len(scope) + 1:]
        x_t, o_t, k = op['x'], op['out'], op['ksize']
        st = e.conv[op['name']]
        xin = read_t(e, x_t)
        w = q(params0[name + '/weights'])
        if op['bn']:
            y = q(R.conv2d(xin, w)) if e._first_fused(op) else read_t(e, op['y'])      # (image layer: recomputed, never stored)
            mean, var = st['mean'].cpu().numpy(), st['var'].cpu().numpy()
            y64 = y.reshape(-1, y.shape[-1]).astype(np.float64)
            assert np.abs(mean - y64.mean(0)).max() <= 2e-5 * np.sqrt(y64.var(0)).max() + 1e-6, name       # moments of the STORED output
            assert np.abs(var - y64.var(0)).max() <= 1e-4 * y64.var(0).max(), name
            gname = name + '/BatchNorm/gamma'
            bname = name + ('/BatchNorm/beta' if (name + '/BatchNorm/beta') in params0 else '/biases')
            z = R.bn_apply(y, mean, var, params0[gname], params0[bname])
            pool = e.fused_pool.get(o_t)
            if pool is not None:
                # gradient arrives at the pooled resolution; routed through the arg-max the forward kernel stored
                dP = read_t(e, pool['out'], grad=True)
                idx = st['pool_idx'].cpu().numpy().reshape(B, o_t.h // 2, o_t.w // 2, o_t.c)
                a = q(R.leaky_relu(z))
                win = np.stack([a[:, dy::2, dx::2, :] for dy in range(2) for dx in range(2)], 0)
                mism = float(np.mean(win.argmax(0) != idx))
                assert mism <= 1e-4, '%s: pool arg-max differs from the oracle on %.2e of the windows' % (name, mism)
                dA = np.zeros_like(z)
                for pos in range(4):
                    dA[:, pos // 2::2, pos % 2::2, :] = np.where(idx == pos, dP, np.float32(0))
            else:
                dA = read_t(e, o_t, grad=True)
            dz = R.leaky_relu_grad(z, dA)
            dy, dg, db = R.bn_train_bwd(y, mean, var, params0[gname], dz)
            dy = q(dy)
            for nm, ref in ((gname, dg), (bname, db)):
                r = rel(grads[nm], ref)
                report.append((r, nm))
                assert r <= tol_vec, '%s rel err %.3e' % (nm, r)
        else:
            dy = read_t(e, o_t, grad=True)
            db = dy.reshape(-1, dy.shape[-1]).astype(np.float64).sum(0)
            r = rel(grads[name + '/biases'], db)
            report.append((r, name + '/biases'))
            assert r <= tol_vec, (name, r)
        dw_ref = R.conv2d_wgrad(xin, dy, k, k)
        r = rel_l2(grads[name + '/weights'], dw_ref)
        report.append((r, name + '/weights'))
        assert r <= tol_l2, '%s/weights rel-L2 %.3e' % (name, r)
        assert rel(grads[name + '/weights'], dw_ref) <= 10 * tol_l2, name
        if x_t not in inputs:
            dx_ref = q(R.conv2d_dgrad(dy, w))
            dx = read_t(e, x_t, grad=True)
            r = rel_l2(dx, dx_ref)
            report.append((r, name + ' dX'))
            assert r <= tol_l2, '%s dX rel-L2 %.3e' % (name, r)
            assert rel(dx, dx_ref) <= (1e-5 if f32 else 1.6e-2), '%s dX max err %.3e' % (name, rel(dx, dx_ref))
    report.sort(reverse=True)
    print('\n%s %d %s B%d C%d teacher-forced backward, worst: %s' % (inference, size, dtype, B, classes, ['%s %.1e' % (n, r) for r, n in report[:5]]))

    # ---- the optimizer on the GPU's own gradient: TF-1.0 ApplyAdam, numerically (not by sign)
    g_dev = {k: v.copy() for k, v in grads.items()}
    sess.apply_gradients()
    params1 = strip(e.get_variables(), scope)
    for k in R.trainable_names(params0):
        w_ref, _, _ = R.adam_step(params0[k], g_dev[k], np.zeros_like(params0[k]), np.zeros_like(params0[k]), lr, 1)
        err = np.abs(params1[k] - w_ref)
        assert np.all(err <= 2.4e-7 * np.abs(w_ref) + 1e-4 * lr), (k, float(err.max()))


def test_multi_scale_training_matches_oracle_per_size(basedir):
    """BASELINE configs[3] (multi-scale {320..608}; the reference lists it as future work, README.md:87): ONE session, buffers
    allocated for 608x608, per-step input size switched with set_size.  At 352 and 320 (cells 11 / 10, reorg input 22 / 20) the
    forward, the loss and the gradients match the oracle on the same weights; 608 runs the largest binding; returning to a size
    reproduces its first result (weights untouched in between: forward_backward only)."""
    from yolo_tf_amd.session import TrainSession
    from yolo_tf_amd.utils import data
    classes, B = 20, 1
    b, _ = make_builder('darknet', classes, 320, True, basedir)
    sess = TrainSession(b, B, dtype='f32', optimizer='adam', learning_rate=1e-3, seed=4, sizes=[(352, 352), (608, 608), (320, 320)])
    e = sess.engine
    scope = 'yolo2_darknet'
    params0 = strip(e.get_variables(), scope)
    rng = np.random.RandomState(8)
    for k in list(params0):
        if k.endswith('gamma'):
            params0[k] = (rng.rand(*params0[k].shape) + 0.5).astype(np.float32)
        if k.endswith(('beta', 'biases')):
            params0[k] = (rng.randn(*params0[k].shape) * 0.1).astype(np.float32)
    e.set_variables({scope + '/' + k: v for k, v in params0.items()})
    spec = R.darknet_spec(classes, len(b.anchors))
    first = {}
    for size in (352, 320, 608, 352, 320):
        cells = size // 32
        sess.set_size(size, size)
        assert (sess.model.cell_width, sess.model.cell_height) == (cells, cells)
        irng = np.random.RandomState(size)
        images = irng.uniform(0, 255, (B, size, size, 3)).astype(np.float32)
        labels = data.synthetic_batch(B, classes, cells, cells, seed=size)
        sess.upload_labels(labels)
        sess.forward_backward(torch.from_numpy(images).cuda())
        got = sess.fetch()
        out = e.output()
        assert (out.h, out.w) == (cells, cells)
        logits = read_t(e, out)
        grads = strip(e.get_gradients(), scope)
        assert torch.all(e.act[out][0].float().reshape(-1, 128)[:B * cells * cells, 125:] == 0)          # padding lanes stay zero
        if size in first:
            # (the 26x26 / 52x52 stages accumulate statistics and filter gradients with f32 atomics: order-dependent last bits)
            # (a last-bit difference in a batch moment can move one activation across the leaky kink / a pool arg-max: allow for that; a
            #  binding that picked up another size's buffers or plans is O(1))
            assert abs(got['total_loss'] - first[size][0]) <= 1e-4 * abs(first[size][0])
            assert rel(logits, first[size][1]) <= 1e-3
            continue
        first[size] = (got['total_loss'], logits.copy())
        x = np.stack([R.per_image_standardization(i) for i in images]).astype(np.float32)
        net, caches = R.network_forward(spec, params0, x, training=True)
        m = R.model_decode(net, classes, b.anchors, training=True)
        obj, aux = R.objectives(m, labels)
        loss = float(R.total_loss(obj, HP))
        assert rel(logits, net) <= 1e-4, (size, rel(logits, net))
        assert abs(got['total_loss'] - loss) <= 1e-4 * abs(loss), (size, got['total_loss'], loss)
        if size != 608:          # (the 608 oracle backward is ~20 s of host time for no new code path)
            ref = R.network_backward(spec, params0, caches, R.loss_backward(m, labels, aux, HP, classes))
            cs = min(cosine(grads[k], ref[k]) for k in grads)
            assert cs >= 0.9995, (size, cs)


def test_multi_scale_bf16_training_steps(basedir):
    """The benchmarked form of configs[3]: bf16, batch 8, a different input size every step, one set of weights: the loss is
    finite at every size and the moving statistics / weights keep evolving through the switches."""
    from yolo_tf_amd.session import TrainSession
    from yolo_tf_amd.utils import data
    B, classes = 8, 20
    sizes = [320, 416, 608, 352, 544]
    b, _ = make_builder('darknet', classes, 416, True, basedir)
    sess = TrainSession(b, B, dtype='bf16', optimizer='adam', learning_rate=1e-4, seed=1, sizes=[(s, s) for s in sizes])
    g = torch.Generator(device='cuda').manual_seed(3)
    before = sess.engine.params.clone()
    for it, size in enumerate(sizes * 2):
        sess.set_size(size, size)
        sess.upload_labels(data.synthetic_batch(B, classes, size // 32, size // 32, seed=it))
        sess.step(torch.rand(B, size, size, 3, device='cuda', generator=g) * 255)
        loss = sess.fetch()['total_loss']
        assert np.isfinite(loss) and 0 < loss < 10, (size, loss)
    assert sess.global_step == 2 * len(sizes)
    assert torch.isfinite(sess.engine.params).all() and float((sess.engine.params - before).abs().max()) > 0


def test_multi_scale_every_size_of_configs3(basedir):
    """All ten input sizes of BASELINE configs[3] ({320, 352, ..., 608}: kernel variants are chosen per shape, so every size takes its
    own plans), not only the five the oracle tests touch.  At 384 / 448 / 480 / 512 / 576 every layer of the bf16 forward is checked against
    the oracle on the engine's own stored inputs (forward_teacher_forced).  Per size, on shared weights: the bf16 engine's logits agree with the f32
    engine's (whose arithmetic is oracle-checked at 320 / 352 / 608) within the whole-network bf16 band, the training loss is finite and equal between the
    two dtypes within 3 %, padding lanes stay zero, gradients are finite, and a launch plan was recorded for the last convolution."""
    from yolo_tf_amd import ops
    from yolo_tf_amd.session import TrainSession
    from yolo_tf_amd.utils import data
    B, classes = 2, 20
    sizes = list(range(320, 609, 32))
    b, _ = make_builder('darknet', classes, 608, True, basedir)
    sess = {dt: TrainSession(b, B, dtype=dt, optimizer='adam', learning_rate=1e-4, seed=5, sizes=[(s_, s_) for s_ in sizes]) for dt in ('f32', 'bf16')}
    g = torch.Generator(device='cuda').manual_seed(7)
    for size in sizes:
        cells = size // 32
        images = torch.rand(B, size, size, 3, device='cuda', generator=g) * 255
        labels = data.synthetic_batch(B, classes, cells, cells, seed=size)
        res = {}
        for dt, se in sess.items():
            se.set_size(size, size)
            se.upload_labels(labels)
            se.forward_backward(images)
            plan = ops.last_conv_plan()
            loss = se.fetch()['total_loss']
            e = se.engine
            out = e.output()
            assert (out.h, out.w) == (cells, cells)
            logits = read_t(e, out)
            pad = e.act[out][0].float().reshape(-1, 128)[:B * cells * cells, 125:]
            assert torch.all(pad == 0), (size, dt)
            assert np.isfinite(loss) and 0 < loss < 10, (size, dt, loss)
            assert torch.isfinite(e.grads).all(), (size, dt)
            assert plan['BM'] in (-1, 128, 256) and plan['grid_x'] != 0, (size, dt, plan)
            res[dt] = (loss, logits)
            if dt == 'bf16' and size in (384, 448, 480, 512, 576):      # the sizes no oracle test touches: every layer's forward vs the oracle
                worst = forward_teacher_forced(e, 'yolo2_darknet', strip(e.get_variables(), 'yolo2_darknet'), False)
                print('\n%d bf16 teacher-forced forward, worst rel-L2: %s' % (size, ['%s %.1e' % (n, r) for r, n in worst[:3]]))
        # (22 layers of bf16 storage on 100-361 samples per channel: the same band as the bf16-vs-oracle network tests; a wrong tile or tap is O(1))
        assert rel_l2(res['bf16'][1], res['f32'][1]) <= 0.25 and rel(res['bf16'][1], res['f32'][1]) <= 0.5, (size, rel_l2(res['bf16'][1], res['f32'][1]), rel(res['bf16'][1], res['f32'][1]))
        assert abs(res['bf16'][0] - res['f32'][0]) <= 3e-2 * abs(res['f32'][0]), (size, res['bf16'][0], res['f32'][0])


def test_multi_scale_batch8_teacher_forced_forward(basedir):
    """BASELINE configs[3] at ITS per-GPU batch (8 images): launch plans depend on M = B H W, so the five sizes the oracle tests only
    touch at batch 2 (384 / 448 / 480 / 512 / 576) are run here at batch 8, bf16, in one multi-size session, and every layer of the training
    forward is checked against the oracle on the engine's own stored inputs; the loss is finite, the padding lanes of the head stay zero."""
    from yolo_tf_amd.session import TrainSession
    from yolo_tf_amd.utils import data
    B, classes = 8, 20
    sizes = [384, 448, 480, 512, 576]
    b, _ = make_builder('darknet', classes, 576, True, basedir)
    sess = TrainSession(b, B, dtype='bf16', optimizer='adam', learning_rate=1e-4, seed=6, sizes=[(s_, s_) for s_ in sizes])
    e = sess.engine
    g = torch.Generator(device='cuda').manual_seed(12)
    for size in sizes:
        cells = size // 32
        sess.set_size(size, size)
        sess.upload_labels(data.synthetic_batch(B, classes, cells, cells, seed=size))
        sess.forward_backward(torch.rand(B, size, size, 3, device='cuda', generator=g) * 255)
        loss = sess.fetch()['total_loss']
        assert np.isfinite(loss) and 0 < loss < 10, (size, loss)
        out = e.output()
        assert (out.h, out.w) == (cells, cells)
        assert torch.all(e.act[out][0].float().reshape(-1, 128)[:B * cells * cells, 125:] == 0), size
        worst = forward_teacher_forced(e, 'yolo2_darknet', strip(e.get_variables(), 'yolo2_darknet'), False)
        print('\n%d batch 8 bf16 teacher-forced forward, worst rel-L2: %s' % (size, ['%s %.1e' % (n, r) for r, n in worst[:3]]))


def test_tensorflow_checkpoint_and_event_file_round_trip(basedir, tmp_path):
    """a training session saved as a TensorFlow V2 checkpoint (the reference's tf.train.Saver layout: variables by TF
    scope name, global_step, <var>/Adam and <var>/Adam_1 slots) restores into a fresh session bit for bit, continues identically,
    and feeds a DetectSession; the summaries land in a TensorBoard event file."""
    from yolo_tf_amd import tf_checkpoint
    from yolo_tf_amd.session import DetectSession, TrainSession
    from yolo_tf_amd.utils import data, events
    os.environ['YOLO2_FUSE_BN_STATS'] = '0'              # two runs are compared bit for bit: order-independent statistics
    try:
        b, _ = make_builder('tiny', 20, 96, True, basedir)
        images = torch.rand(2, 96, 96, 3, device='cuda', generator=torch.Generator(device='cuda').manual_seed(0)) * 255
        labels = data.synthetic_batch(2, 20, 3, 3, seed=1)

        def fresh():
            return TrainSession(b, 2, dtype='f32', optimizer='adam', learning_rate=1e-3, seed=9)
        a = fresh()
        writer = events.FileWriter(str(tmp_path / 'run'))
        for _ in range(3):
            a.step(images, labels)
            writer.add_training_summary(a.global_step, a.fetch())
        writer.close()
        prefix = tf_checkpoint.save(str(tmp_path), a)
        assert tf_checkpoint.latest_checkpoint(str(tmp_path)) == prefix and a.global_step == 3
        names = set(tf_checkpoint.read_index(prefix)) - {''}
        assert 'global_step' in names and 'yolo2_tiny/conv0/weights' in names and 'yolo2_tiny/conv0/weights/Adam_1' in names
        c = fresh()
        assert tf_checkpoint.restore(prefix, c) == 3
        assert torch.equal(c.engine.params, a.engine.params) and torch.equal(c.engine.state, a.engine.state)
        assert all(torch.equal(x, y) for x, y in zip(c.optimizer.slots, a.optimizer.slots))
        a.step(images, labels)
        c.step(images, labels)
        torch.cuda.synchronize()
        assert c.global_step == 4
        # Filter gradients of split pixel ranges accumulate with f32 atomics, in an order that depends on the buffers' addresses: the two sessions'
        # gradients are equal to rounding, not bitwise.  Adam divides by sqrt(v): a parameter whose gradient is a cancelling sum at the noise floor
        # in all four steps can move by a visible fraction of the step size (1e-3) in either session -- one such element failed a max <= 1e-6 bound
        # once in some ten whole-suite runs (25 isolated repeats: max 1.2e-7).  So: all but a vanishing share agree to 1e-6, none by more than a step.
        dpar = (c.engine.params - a.engine.params).abs()
        print('\ncontinued step: max |dparam| %.3e, share above 1e-6: %.2e' % (float(dpar.max()), float((dpar > 1e-6).float().mean())))
        assert float((dpar > 1e-6).float().mean()) <= 1e-5 and float(dpar.max()) <= 2.5e-3
        ev = events.read_events(writer.path)
        assert [e['step'] for e in ev] == [0, 1, 2, 3] and [t for t, _ in ev[1]['scalars']] == list(events.SCALAR_TAGS)
        bd, _ = make_builder('tiny', 20, 96, False, basedir)
        det = DetectSession(bd, 1, dtype='f32')
        assert tf_checkpoint.restore(prefix, engine=det.engine) == 3
        got = det.engine.get_variables()
        ref = a.engine.get_variables()
        ref3 = tf_checkpoint.read(prefix)
        assert all(np.array_equal(got[k], ref3[k]) for k in got) and set(got) <= set(ref)
    finally:
        del os.environ['YOLO2_FUSE_BN_STATS']


@pytest.mark.parametrize('B,sizes', [(16, [416]), (8, [416]), (4, [416]), (8, [320, 384, 480, 544, 608])])
def test_no_layer_falls_back_to_separate_finalisation(basedir, B, sizes, monkeypatch):
    """Every batch-normalised layer of a Darknet-19 training step takes the consumers that finish the statistics / dgamma, dbeta sums in
    their own prologue -- at the bench's batch AND at the smaller per-GPU batches of the strong-scaling split and the multi-scale sizes
    (at batch 8 the 26x26 stage once produced more unique partial rows than its consumer reads and silently took the two-launch form)."""
    from yolo_tf_amd import ops
    from yolo_tf_amd.session import TrainSession
    from yolo_tf_amd.utils import data
    if os.environ.get('YOLO2_FOLD_FINALIZE', '1') == '0' or os.environ.get('YOLO2_FUSE_BN_STATS', '1') == '0':
        pytest.skip('folded finalisation switched off')
    b, _ = make_builder('darknet', 20, max(sizes), True, basedir)
    sess = TrainSession(b, B, dtype='bf16', optimizer='adam', learning_rate=1e-4, seed=3, sizes=[(s_, s_) for s_ in sizes] if len(sizes) > 1 else None)
    calls = []
    for name in ('bn_finalize', 'bn_part_to_grads', 'bn_leaky', 'bn_leaky_pool', 'bn_leaky_bwd_apply', 'bn_leaky_pool_bwd_apply', 'bn_stats_ema'):
        orig = getattr(ops, name)
        monkeypatch.setattr(ops, name, (lambda o, n: (lambda *a, **k: (calls.append(n), o(*a, **k))[1]))(orig, name))
    g = torch.Generator(device='cuda').manual_seed(1)
    for size in sizes:
        if len(sizes) > 1:
            sess.set_size(size, size)
        sess.upload_labels(data.synthetic_batch(B, 20, size // 32, size // 32, seed=size))
        sess.step(torch.rand(B, size, size, 3, device='cuda', generator=g) * 255)
        torch.cuda.synchronize()
        # (fewer pixels in the last stage than the batch-8 bench shape: its 1x1 layers run K-sliced data gradients, which have no on-chip
        #  tile to take the producer's sums from; the library then runs reduce + finalise itself and the plain apply pass follows -- by design)
        allowed = {'bn_leaky_bwd_apply'} if B * (size // 32) ** 2 < 8 * 13 * 13 else set()
        assert set(calls) <= allowed, (B, size, sorted(set(calls)))
        del calls[:]


def _sync_bn_worker(rank, world, port, outdir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    import torch.distributed as dist
    from yolo_tf_amd.parallel import init_distributed
    from yolo_tf_amd.session import TrainSession
    torch.cuda.set_device(0)
    init_distributed(backend='gloo')                    # both ranks share the one GPU of the test box: gloo carries the CUDA tensors
    d = np.load(os.path.join(outdir, 'data.npz'))
    b, _ = make_builder('darknet', 20, 96, True, os.path.join(outdir, 'base%d' % rank))
    sess = TrainSession(b, 2, dtype='f32', optimizer='adam', learning_rate=1e-3, seed=3, world_size=world, bucket_mb=8.0, sync_bn=True)
    e = sess.engine
    assert e.sync_bn and e.bn_world == 2 and not e.fold_finalize
    sess.upload_labels([d['l%d' % i][2 * rank:2 * rank + 2] for i in range(6)])
    images = torch.from_numpy(d['images'][2 * rank:2 * rank + 2]).cuda()
    sess.step(images)
    torch.cuda.synchronize()
    stats = {k: np.asarray(v, np.float32) for k, v in e.get_variables().items() if 'moving_' in k}
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), grads=e.grads.cpu().numpy(), params=e.params.cpu().numpy(), logits=read_t(e, e.output()),
             loss=np.float64(sess.fetch()['total_loss']), **{'mv/' + k: v for k, v in stats.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_sync_bn_two_ranks_equal_one_process_with_the_joint_batch(tmp_path, basedir):
    """[mi355x] sync_bn: two ranks x 2 images with batch moments and BN-backward sums exchanged train exactly like ONE process on the four
    images -- logits, averaged gradients, moving statistics and the updated parameters agree to f32 rounding.
    (The default -- replica-local statistics -- is what two independent reference processes would compute.)"""
    import socket
    import torch.multiprocessing as mp
    from yolo_tf_amd.session import TrainSession
    from yolo_tf_amd.utils import data
    rng = np.random.RandomState(11)
    images = rng.uniform(0, 255, (4, 96, 96, 3)).astype(np.float32)
    labels = data.synthetic_batch(4, 20, 3, 3, seed=12)
    np.savez(str(tmp_path / 'data.npz'), images=images, **{'l%d' % i: np.asarray(l) for i, l in enumerate(labels)})
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_sync_bn_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(str(tmp_path / 'rank0.npz')), np.load(str(tmp_path / 'rank1.npz'))
    # the single process on the joint batch (two-launch finalisation, like the synchronised run)
    b, _ = make_builder('darknet', 20, 96, True, basedir)
    sess = TrainSession(b, 4, dtype='f32', optimizer='adam', learning_rate=1e-3, seed=3)
    sess.upload_labels(labels)
    sess.forward_backward(torch.from_numpy(images).cuda())
    e = sess.engine
    joint_g = e.grads.cpu().numpy()
    joint_logits = read_t(e, e.output())
    joint_loss = sess.fetch()['total_loss']
    sess.apply_gradients()
    torch.cuda.synchronize()
    both_logits = np.concatenate([r0['logits'], r1['logits']])
    assert rel_l2(both_logits, joint_logits) <= 1e-4, rel_l2(both_logits, joint_logits)
    assert abs(0.5 * (float(r0['loss']) + float(r1['loss'])) - joint_loss) <= 1e-5 * abs(joint_loss)
    np.testing.assert_array_equal(r0['grads'], r1['grads'])            # the exchanged (summed) gradient arena
    g_avg = r0['grads'] / 2.0                                          # sum over ranks of per-rank-mean losses / world == gradient of the joint mean
    # (36 samples per channel in the last stage: one activation on the other side of the leaky kink or another pool arg-max -- f32 atomics
    #  order the sums differently in every run -- moves the whole gradient by ~1e-2; replica-local statistics would move it by O(1))
    assert rel_l2(g_avg, joint_g) <= 5e-2, rel_l2(g_avg, joint_g)
    per_tensor = {name: cosine(g_avg[o:o + n], joint_g[o:o + n]) for name, (o, n) in e.param_offsets.items() if np.abs(joint_g[o:o + n]).max() > 0}
    assert np.median(list(per_tensor.values())) >= 0.999, sorted(per_tensor.items(), key=lambda kv: kv[1])[:3]
    np.testing.assert_array_equal(r0['params'], r1['params'])
    for k, v in e.get_variables().items():
        if 'moving_' in k:
            assert np.abs(r0['mv/' + k] - np.asarray(v, np.float32)).max() <= 1e-6 + 1e-5 * np.abs(r0['mv/' + k]).max(), k
            np.testing.assert_array_equal(r0['mv/' + k], r1['mv/' + k])
    # Adam's first step is +-alpha whatever the magnitude: compare the direction of the update instead of its size
    p_joint = e.params.cpu().numpy()
    agree = float(np.mean(np.abs(r0['params'] - p_joint) <= 1e-6))     # (a gradient whose sign differs -- |g| at rounding level -- moves by 2 alpha = 2e-3)
    assert agree >= 0.90, agree


def _shard_opt_worker(rank, world, port, outdir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    import torch.distributed as dist
    from yolo_tf_amd.parallel import GradReducer, init_distributed
    from yolo_tf_amd.session import TrainSession
    torch.cuda.set_device(0)
    init_distributed(backend='gloo')                    # both ranks share the one GPU of the test box: gloo carries the CUDA tensors
    d = np.load(os.path.join(outdir, 'data.npz'))
    images = torch.from_numpy(d['images'][2 * rank:2 * rank + 2]).cuda()
    labels = [d['l%d' % i][2 * rank:2 * rank + 2] for i in range(6)]
    out = {}
    # (a) the exchange + update on IDENTICAL local gradients (a second backward would differ in the last bits: f32 atomics): the sharded
    #     chain -- reduce-scatter, Adam on the own shard, all-gather -- against all-reduce + Adam over the whole arena
    b, _ = make_builder('tiny', 20, 96, True, os.path.join(outdir, 'base%d' % rank))
    sess = TrainSession(b, 2, dtype='f32', optimizer='adam', learning_rate=1e-3, seed=3, world_size=world, bucket_mb=4.0, shard_optimizer=True)
    e, opt = sess.engine, sess.optimizer
    assert sess.shard_optimizer and len(sess.reducer.buckets) >= 3
    sess.upload_labels(labels)
    keep = sess.reducer
    sess.reducer = None
    sess.forward_backward(images)                        # local gradients, no collective
    sess.reducer = keep
    torch.cuda.synchronize()
    local_g, p0 = e.grads.clone(), e.params.clone()
    g_sum = local_g.clone()
    dist.all_reduce(g_sum)
    e.grads.copy_(g_sum)
    opt.apply(e.params, e.grads, 1e-3, 1, 1.0 / world)
    torch.cuda.synchronize()
    out['rep/params'], out['rep/m'], out['rep/v'] = e.params.cpu().numpy(), opt.slots[0].cpu().numpy(), opt.slots[1].cpu().numpy()
    e.params.copy_(p0)
    e.grads.copy_(local_g)
    for sl in opt.slots:
        sl.zero_()
    red = GradReducer(e.grads, list(e.param_offsets.values()), 4.0, shard_params=e.params)
    red.update_fn = lambda lo, hi: opt.apply(e.params, e.grads, 1e-3, 1, 1.0 / world, lo, hi)
    red.begin()
    red.finish(wait=True)
    red.gather_slots(opt.slots)
    torch.cuda.synchronize()
    out['sh/params'], out['sh/m'], out['sh/v'] = e.params.cpu().numpy(), opt.slots[0].cpu().numpy(), opt.slots[1].cpu().numpy()
    # (b) the session path: three sharded steps (chains enqueued during backward, update on the communication stream)
    for sl in opt.slots:
        sl.zero_()
    e.params.copy_(p0)
    e._filters_dirty = True
    for step in range(3):
        sess.upload_labels(labels)
        sess.step(images)
    sess.gather_optimizer_state()
    torch.cuda.synchronize()
    assert sess.global_step == 3
    out['steps/params'], out['steps/m'] = e.params.cpu().numpy(), opt.slots[0].cpu().numpy()
    out['p0'] = p0.cpu().numpy()
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


def test_optimizer_sharding_two_ranks_equal_replicated_update(tmp_path):
    """[mi355x] shard_optimizer on the GPU, two ranks.  On identical local gradients the sharded exchange + update leaves parameters and
    (gathered) Adam moments bit-identical to all-reduce + the replicated update; through TrainSession.step the replicas stay bit-identical
    to each other and every parameter with a gradient moves."""
    import socket
    import torch.multiprocessing as mp
    from yolo_tf_amd.utils import data
    rng = np.random.RandomState(21)
    images = rng.uniform(0, 255, (4, 96, 96, 3)).astype(np.float32)
    labels = data.synthetic_batch(4, 20, 3, 3, seed=22)
    np.savez(str(tmp_path / 'data.npz'), images=images, **{'l%d' % i: np.asarray(l) for i, l in enumerate(labels)})
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_shard_opt_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(str(tmp_path / 'rank0.npz')), np.load(str(tmp_path / 'rank1.npz'))
    for k in ('params', 'm', 'v'):
        np.testing.assert_array_equal(r0['sh/' + k], r0['rep/' + k])
        np.testing.assert_array_equal(r1['sh/' + k], r1['rep/' + k])
        np.testing.assert_array_equal(r0['sh/' + k], r1['sh/' + k])
    assert np.abs(r0['sh/m']).max() > 0
    np.testing.assert_array_equal(r0['steps/params'], r1['steps/params'])
    np.testing.assert_array_equal(r0['steps/m'], r1['steps/m'])
    moved = np.abs(r0['steps/params'] - r0['p0']) > 0
    assert moved.mean() > 0.5, moved.mean()              # Adam moves every parameter that has a gradient by ~lr per step

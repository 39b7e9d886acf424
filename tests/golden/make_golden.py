"""Generates tests/golden/*.npz by running the REFERENCE's own NumPy-only functions.

Run once in the build container (needs /root/reference; TensorFlow is absent, so the
``tensorflow`` import chain is stubbed and ``np.int`` -- removed from NumPy -- is shimmed):

    python tests/golden/make_golden.py

Only inputs and expected outputs are stored (data); no reference source travels with the repo.
Reference functions exercised:
    utils/postprocess.py:21-51      iou, non_max_suppress
    utils/data/__init__.py:112-145  transform_labels
    model/yolo/__init__.py:29-34    calc_cell_xy
    utils/preprocess.py:23-25       per_image_standardization
    parse_darknet_yolo2.py:34-48    transpose_weights, transpose_biases
    model/yolo2/function.py:32-47   reorg known-answer image (the KAT's input and its asserted per-channel constants)
and, executed under the NumPy-backed TensorFlow stand-in of tests/golden/tf_numpy_shim.py (which states what such a run
pins -- everything the reference's Python decides -- and what it cannot: the arithmetic inside each elementary TF op):
    model/yolo2/__init__.py:28-94   Model (every attribute) and Objectives (the four terms)       -> model.npz
    model/yolo/__init__.py:37-100   the YOLO (v1) Model and Objectives                            -> model.npz
    model/yolo2/function.py:22-47   reorg on a random tensor; the reference's own main() KAT run  -> model.npz
    model/yolo2/inference.py:25-120 tiny / darknet / _tiny / _darknet: layer tables, variable names, concat order (topology.json)
    model/yolo/inference.py:23-64   and the logits of seeded weights on a seeded image (network.npz; tests/golden/seeded.py)
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import seeded            # noqa: E402
import tf_numpy_shim as shim   # noqa: E402

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub_tf():
    shim.install()           # tensorflow, tensorflow.contrib.slim, tensorflow.python.client.device_lib: NumPy-backed stand-ins
    names = ['matplotlib', 'matplotlib.patches', 'matplotlib.pyplot']
    for n in names:
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
    sys.modules['matplotlib'].patches = sys.modules['matplotlib.patches']
    if not hasattr(np, 'int'):
        np.int = int


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def boxes_from(center, wh):
    center = center.astype(np.float32)
    wh = wh.astype(np.float32)
    return (center - wh / 2).astype(np.float32), (center + wh / 2).astype(np.float32)


def nms_case(post, name, conf, xy_min, xy_max, thr, thr_iou, cases):
    conf_in = conf.copy()
    n = conf.shape[0] * conf.shape[1]
    tag = np.arange(n, dtype=np.float32)
    # carry the original index through the reference's list by appending it to xy_min rows? No --
    # the reference returns views; recover the order by matching row addresses instead.
    work = conf.copy()
    boxes = post.non_max_suppress(work, xy_min, xy_max, thr, thr_iou)
    flat = work.reshape(n, -1)
    base = flat.__array_interface__['data'][0]
    stride = flat.strides[0]
    order = np.array([(b[0].__array_interface__['data'][0] - base) // stride for b in boxes], np.int64)
    assert sorted(order.tolist()) == list(range(n))
    cases[name + '/conf_in'] = conf_in
    cases[name + '/xy_min'] = xy_min
    cases[name + '/xy_max'] = xy_max
    cases[name + '/thr'] = np.float64(thr)
    cases[name + '/thr_iou'] = np.float64(thr_iou)
    cases[name + '/conf_out'] = work
    cases[name + '/order'] = order
    del tag


def make_nms(post):
    cases = {}
    # 1. sparse realistic, 845 x 20 (BASELINE.md section 2 recipe)
    for name, classes, seed in (('sparse20', 20, 1), ('sparse80', 80, 2)):
        rng = np.random.RandomState(seed)
        conf = rng.uniform(0, 0.05, (169, 5, classes)).astype(np.float32)
        hot = rng.choice(845, 12, replace=False)
        conf.reshape(845, classes)[hot, rng.randint(0, classes, 12)] = rng.uniform(0.5, 0.9, 12).astype(np.float32)
        mn, mx = boxes_from(rng.uniform(0, 13, (169, 5, 2)), rng.uniform(0.5, 5.5, (169, 5, 2)))
        nms_case(post, name, conf, mn, mx, 0.3, 0.4, cases)
    # 2. dense random (small enough for the Python reference): 40 cells x 5 anchors x 6 classes
    rng = np.random.RandomState(0)
    conf = rng.uniform(0, 0.5, (40, 5, 6)).astype(np.float32)
    mn, mx = boxes_from(rng.uniform(0, 13, (40, 5, 2)), rng.uniform(0, 4, (40, 5, 2)))
    nms_case(post, 'dense', conf, mn, mx, 0.3, 0.4, cases)
    # 2b. the full-size dense stress case of SURVEY 8c(2): every one of 845 x 20 scores is a candidate (~70 s in the Python reference)
    rng = np.random.RandomState(10)
    conf = rng.uniform(0.3, 1.0, (169, 5, 20)).astype(np.float32)
    mn, mx = boxes_from(rng.uniform(0, 13, (169, 5, 2)), rng.uniform(0.5, 5.5, (169, 5, 2)))
    nms_case(post, 'dense845', conf, mn, mx, 0.3, 0.4, cases)
    # 3. clustered duplicates: 8 clusters of jittered boxes
    rng = np.random.RandomState(3)
    cen = np.repeat(rng.uniform(2, 11, (8, 2)), 20, 0) + rng.normal(0, 0.15, (160, 2))
    wh = np.repeat(rng.uniform(1, 4, (8, 2)), 20, 0) * rng.uniform(0.9, 1.1, (160, 2))
    mn, mx = boxes_from(cen.reshape(32, 5, 2), wh.reshape(32, 5, 2))
    conf = rng.uniform(0, 1, (32, 5, 4)).astype(np.float32)
    nms_case(post, 'clustered', conf, mn, mx, 0.3, 0.4, cases)
    # 4. identical boxes (IoU == 1) and equal-score ties (stable order must carry across classes)
    rng = np.random.RandomState(4)
    mn, mx = boxes_from(np.tile(np.float32([[5, 5]]), (30, 1)).reshape(6, 5, 2), np.tile(np.float32([[2, 3]]), (30, 1)).reshape(6, 5, 2))
    conf = rng.choice(np.float32([0.25, 0.5, 0.5, 0.75]), (6, 5, 3)).astype(np.float32)
    nms_case(post, 'identical_ties', conf, mn, mx, 0.3, 0.4, cases)
    # 5. ties with distinct geometry: scores drawn from 4 values, boxes random
    rng = np.random.RandomState(5)
    conf = rng.choice(np.float32([0.1, 0.35, 0.6, 0.6, 0.9]), (20, 5, 5)).astype(np.float32)
    mn, mx = boxes_from(rng.uniform(0, 13, (20, 5, 2)), rng.uniform(1, 6, (20, 5, 2)))
    nms_case(post, 'ties', conf, mn, mx, 0.3, 0.4, cases)
    # 6. IoU exactly at threshold: unit squares offset so that IoU = 1/3 and 0.5 (>= must fire)
    mn = np.float32([[0, 0], [0.5, 0], [4, 4], [4, 4.5], [8, 8]]).reshape(1, 5, 2)
    mx = mn + np.float32(1)
    conf = np.float32([[0.9], [0.8], [0.7], [0.6], [0.5]]).reshape(1, 5, 1)
    iou_pair = post.iou(mn[0, 0], mx[0, 0], mn[0, 1], mx[0, 1])
    nms_case(post, 'at_threshold', conf, mn, mx, 0.3, float(iou_pair), cases)
    # 7. all below threshold -> no-op
    rng = np.random.RandomState(7)
    conf = rng.uniform(0, 0.29, (10, 5, 3)).astype(np.float32)
    mn, mx = boxes_from(rng.uniform(0, 13, (10, 5, 2)), rng.uniform(1, 6, (10, 5, 2)))
    nms_case(post, 'all_below', conf, mn, mx, 0.3, 0.4, cases)
    # 8. zero-area boxes (union floor 1e-10) mixed with normal ones
    rng = np.random.RandomState(8)
    cen = rng.uniform(0, 13, (10, 5, 2))
    wh = rng.uniform(0, 3, (10, 5, 2))
    wh[::2, :, 0] = 0
    mn, mx = boxes_from(cen, wh)
    conf = rng.uniform(0, 1, (10, 5, 2)).astype(np.float32)
    nms_case(post, 'zero_area', conf, mn, mx, 0.3, 0.4, cases)
    # iou table
    rng = np.random.RandomState(9)
    a_mn, a_mx = boxes_from(rng.uniform(0, 13, (64, 2)), rng.uniform(0, 5, (64, 2)))
    b_mn, b_mx = boxes_from(rng.uniform(0, 13, (64, 2)), rng.uniform(0, 5, (64, 2)))
    cases['iou/a_min'], cases['iou/a_max'], cases['iou/b_min'], cases['iou/b_max'] = a_mn, a_mx, b_mn, b_mx
    cases['iou/out'] = np.array([post.iou(a_mn[i], a_mx[i], b_mn[i], b_mx[i]) for i in range(64)], np.float32)
    np.savez_compressed(os.path.join(OUT, 'nms.npz'), **cases)
    print('nms.npz', len(cases), 'arrays')


def make_labels(data_mod, yolo_mod, pre_mod):
    cases = {}
    for name, classes, cw, ch, k, seed in (('voc13', 20, 13, 13, 6, 0), ('coco13', 80, 13, 13, 9, 1),
                                           ('rect', 20, 19, 10, 5, 2), ('shared_cell', 20, 13, 13, 4, 3)):
        rng = np.random.RandomState(seed)
        cen = rng.uniform(0.05, 0.95, (k, 2))
        wh = rng.uniform(0.05, 0.6, (k, 2))
        if name == 'shared_cell':
            cen[1] = cen[0] + 0.001  # two objects in one cell -> multi-hot prob, last one wins elsewhere
        coord = np.clip(np.concatenate([cen - wh / 2, cen + wh / 2], 1), 0, 1).astype(np.float32)
        cls = rng.randint(0, classes, k).astype(np.int64)
        out = data_mod.transform_labels(cls, coord, classes, cw, ch)
        cases[name + '/class'] = cls
        cases[name + '/coord'] = coord
        cases[name + '/dims'] = np.array([classes, cw, ch], np.int64)
        for key, v in zip(('mask', 'prob', 'coords', 'offset_xy_min', 'offset_xy_max', 'areas'), out):
            cases[name + '/' + key] = v
    cases['cell_xy/13x13'] = yolo_mod.calc_cell_xy(13, 13)
    cases['cell_xy/10x19'] = yolo_mod.calc_cell_xy(10, 19)
    rng = np.random.RandomState(11)
    img = rng.uniform(0, 255, (32, 48, 3)).astype(np.float32)
    cases['std/in'] = img
    cases['std/out'] = pre_mod.per_image_standardization(img)
    flat = np.full((8, 8, 3), 7, np.float32)
    cases['std/flat_in'] = flat
    cases['std/flat_out'] = pre_mod.per_image_standardization(flat)
    # reorg KAT (model/yolo2/function.py:33-47): input image and the asserted channel constants
    image = np.array([(0, 1, 0, 1), (2, 3, 2, 3), (0, 1, 0, 1), (2, 3, 2, 3)], np.uint8)[None, :, :, None]
    cases['reorg/kat_in'] = image
    cases['reorg/kat_channel_values'] = np.arange(4, dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, 'labels.npz'), **cases)
    print('labels.npz', len(cases), 'arrays')


def make_weights():
    """parse_darknet_yolo2.py:34-48 transpose_weights / transpose_biases (final-layer channel permutation)."""
    for n in ('pandas',):
        try:
            __import__(n)
        except ImportError:
            sys.modules[n] = types.ModuleType(n)
    ref = _load('parse_darknet_yolo2.py', 'ref_parse_darknet')
    rng = np.random.RandomState(21)
    cases = {}
    for name, anchors, classes, cin in (('voc', 5, 20, 7), ('coco', 5, 80, 3), ('a3c4', 3, 4, 2)):
        w = rng.randn(1, 1, cin, anchors * (5 + classes)).astype(np.float32)
        b = rng.randn(anchors * (5 + classes)).astype(np.float32)
        cases[name + '/anchors'] = np.int64(anchors)
        cases[name + '/w_in'], cases[name + '/b_in'] = w, b
        cases[name + '/w_out'] = ref.transpose_weights(w, anchors)
        cases[name + '/b_out'] = ref.transpose_biases(b, anchors)
    np.savez_compressed(os.path.join(OUT, 'weights.npz'), **cases)
    print('weights.npz', len(cases), 'arrays')


def _labels_for(ref_data, classes, cw, ch, batch, seed):
    """Label tensors produced by the reference's own transform_labels for seeded boxes, stacked over the batch."""
    rng = np.random.RandomState(seed)
    per_image = []
    for _ in range(batch):
        k = rng.randint(1, 7)
        cen = rng.uniform(0.05, 0.95, (k, 2))
        wh = rng.uniform(0.05, 0.6, (k, 2))
        coord = np.clip(np.concatenate([cen - wh / 2, cen + wh / 2], 1), 0, 1).astype(np.float32)
        cls = rng.randint(0, classes, k).astype(np.int64)
        per_image.append(ref_data.transform_labels(cls, coord, classes, cw, ch))
    return [np.stack([p[i] for p in per_image]).astype(np.float32) for i in range(6)]


MODEL_ATTRS = ('iou', 'offset_xy', 'wh', 'prob', 'areas', 'offset_xy_min', 'offset_xy_max', 'wh01', 'wh01_sqrt', 'coords', 'xy', 'xy_min', 'xy_max', 'conf')
MODEL1_ATTRS = ('prob', 'iou', 'offset_xy', 'coords', 'wh', 'offset_xy_min', 'offset_xy_max', 'areas', 'xy', 'xy_min', 'xy_max', 'conf')
LABEL_KEYS = ('mask', 'prob', 'coords', 'offset_xy_min', 'offset_xy_max', 'areas')


def make_model(ref_yolo2, ref_yolo, ref_fn, ref_data):
    """The reference's Model / Objectives classes and reorg, executed on seeded logits and on labels from its transform_labels."""
    import tensorflow as tf
    cases = {}
    here = os.path.join(os.path.dirname(OUT), '..', 'config', 'yolo2', 'anchors')
    for name, classes, tsv, ch, cw, batch, seed in (('voc13', 20, 'voc.tsv', 13, 13, 2, 31), ('coco_rect', 80, 'coco.tsv', 10, 19, 1, 32),
                                                    ('voc_big_logits', 20, 'voc.tsv', 13, 13, 2, 33)):
        shim.reset()
        anchors = np.loadtxt(os.path.join(here, tsv), delimiter='\t', skiprows=1)          # float64, as pandas read_csv(...).values gives the reference
        rng = np.random.RandomState(seed)
        scale = 4.0 if name == 'voc_big_logits' else 1.0
        net = (rng.standard_normal((batch, ch, cw, len(anchors) * (5 + classes))) * scale).astype(np.float32)
        labels = _labels_for(ref_data, classes, cw, ch, batch, seed + 100)
        model = ref_yolo2.Model(tf.constant(net), classes, anchors, training=False)
        obj = ref_yolo2.Objectives(model, *[tf.constant(a) for a in labels])
        cases[name + '/net'], cases[name + '/anchors'], cases[name + '/classes'] = net, anchors, np.int64(classes)
        for k, v in zip(LABEL_KEYS, labels):
            cases[name + '/labels/' + k] = v
        for k in MODEL_ATTRS:
            cases[name + '/model/' + k] = getattr(model, k).value
        assert list(obj.keys()) == ['iou_best', 'iou_normal', 'coords', 'prob']
        for k in obj:
            cases[name + '/objectives/' + k] = obj[k].value
    # YOLO (v1): per-cell class scores + boxes_per_cell x (iou, xy, sqrt wh)
    for name, classes, boxes, ch, cw, batch, seed in (('v1_voc7', 20, 2, 7, 7, 2, 41), ('v1_rect', 4, 3, 3, 5, 1, 42)):
        shim.reset()
        rng = np.random.RandomState(seed)
        tf.identity(tf.constant(np.zeros((batch, ch, cw, 8), np.float32)), name='yolo_tiny/conv')      # the tensor Model reads the grid size from (:39)
        net = rng.uniform(-0.2, 1.0, (batch, ch * cw * (classes + boxes * 5))).astype(np.float32)
        labels = _labels_for(ref_data, classes, cw, ch, batch, seed + 100)
        model = ref_yolo.Model(tf.constant(net), 'yolo_tiny', classes, boxes, training=False)
        obj = ref_yolo.Objectives(model, *[tf.constant(a) for a in labels])
        cases[name + '/net'], cases[name + '/dims'] = net, np.array([classes, boxes, ch, cw], np.int64)
        for k, v in zip(LABEL_KEYS, labels):
            cases[name + '/labels/' + k] = v
        for k in MODEL1_ATTRS:
            cases[name + '/model/' + k] = getattr(model, k).value
        for k in obj:
            cases[name + '/objectives/' + k] = obj[k].value
    # reorg: the reference's own known-answer main() must pass under the stand-in, then a random tensor
    shim.reset()
    ref_fn.main()
    rng = np.random.RandomState(51)
    x = rng.standard_normal((2, 26, 26, 8)).astype(np.float32)
    cases['reorg/in'] = x
    cases['reorg/out'] = ref_fn.reorg(tf.constant(x)).value
    np.savez_compressed(os.path.join(OUT, 'model.npz'), **cases)
    print('model.npz', len(cases), 'arrays')


def make_topology(ref_inf2, ref_inf1):
    """Layer tables + logits of the reference's inference functions (slim calls recorded and evaluated by the stand-in)."""
    import tensorflow as tf
    topo, cases = {}, {}
    runs = (('yolo2_darknet', ref_inf2.darknet, 20, 5, 64), ('yolo2_tiny', ref_inf2.tiny, 20, 5, 64),
            ('yolo2__darknet', ref_inf2._darknet, 20, 5, 64), ('yolo2__tiny', ref_inf2._tiny, 20, 5, 64),
            ('yolo2_darknet_coco', ref_inf2.darknet, 80, 5, 64), ('yolo_tiny', ref_inf1.tiny, 20, 2, 128))
    for key, fn, classes, boxes, size in runs:
        rng = np.random.RandomState(61)
        image = rng.standard_normal((2, size, size, 3)).astype(np.float32)
        entry = {'function': fn.__name__, 'classes': classes, 'boxes': boxes, 'input': [2, size, size, 3]}
        for training in (False, True):
            if training and key in ('yolo_tiny', 'yolo2__darknet', 'yolo2__tiny', 'yolo2_darknet_coco'):
                continue                              # (v1: dropout draws; the variants: same batch-norm code path as their base)
            shim.reset(lambda name, shape, kind: seeded.value(name, shape, kind))
            scope, net = fn(tf.constant(image), classes, boxes, training=training)
            tag = key + ('/train' if training else '/infer')
            cases[tag + '/logits'] = net.value
            if training:
                names = sorted(shim.UPDATES)
                for n in (names[0], names[1], names[-2], names[-1]):
                    cases[tag + '/update/' + n] = shim.UPDATES[n]
            else:
                entry['scope'] = scope
                entry['layers'] = list(shim.LOG)
                entry['variables'] = list(shim.VAR_INFO)
                cases[key + '/image'] = image
        consts = {k: list(getattr(ref_inf1 if key == 'yolo_tiny' else ref_inf2, k)) for k in dir(ref_inf1 if key == 'yolo_tiny' else ref_inf2) if k.endswith('_DOWNSAMPLING')}
        entry['downsampling'] = consts
        topo[key] = entry
    # Training-mode (batch statistics) logits on 128x128 images: 2 x 4 x 4 = 32 samples per channel in the 1/32 stages, 128 at 1/16, ...
    # (the 64x64 case above leaves 8 there, where a batch variance is mostly noise); COCO-80 head; and `_darknet` -- the
    # biases-instead-of-beta branch (model/yolo2/inference.py:122-126 calls darknet(..., center=False))
    for key, base, fn, classes, size in (('yolo2_darknet_t128', 'yolo2_darknet', ref_inf2.darknet, 20, 128), ('yolo2_darknet_coco_t128', 'yolo2_darknet_coco', ref_inf2.darknet, 80, 128),
                                         ('yolo2__darknet_t128', 'yolo2__darknet', ref_inf2._darknet, 20, 128)):
        rng = np.random.RandomState(62)
        image = rng.standard_normal((2, size, size, 3)).astype(np.float32)
        shim.reset(lambda name, shape, kind: seeded.value(name, shape, kind))
        scope, net = fn(tf.constant(image), classes, 5, training=True)
        cases[key + '/image'] = image
        cases[key + '/train/logits'] = net.value
        names = sorted(shim.UPDATES)
        for n in (names[0], names[1], names[-2], names[-1]):
            cases[key + '/train/update/' + n] = shim.UPDATES[n]
        topo[key] = {'base': base, 'function': fn.__name__, 'classes': classes, 'boxes': 5, 'input': [2, size, size, 3], 'scope': scope}
    with open(os.path.join(OUT, 'topology.json'), 'w') as f:
        json.dump(topo, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, 'network.npz'), **cases)
    print('topology.json', len(topo), 'networks; network.npz', len(cases), 'arrays')


def main():
    _stub_tf()
    sys.path.insert(0, REF)
    post = _load('utils/postprocess.py', 'ref_postprocess')
    pre = _load('utils/preprocess.py', 'ref_preprocess')
    import utils.data as ref_data          # noqa: E402  (reference package, stubbed TF)
    import model.yolo as ref_yolo          # noqa: E402
    import model.yolo2 as ref_yolo2        # noqa: E402
    import model.yolo2.function as ref_fn  # noqa: E402
    import model.yolo2.inference as ref_inf2   # noqa: E402
    import model.yolo.inference as ref_inf1    # noqa: E402
    only = sys.argv[1:]
    if not only or 'nms' in only:
        make_nms(post)
    if not only or 'labels' in only:
        make_labels(ref_data, ref_yolo, pre)
    if not only or 'weights' in only:
        make_weights()
    if not only or 'model' in only:
        make_model(ref_yolo2, ref_yolo, ref_fn, ref_data)
    if not only or 'topology' in only:
        make_topology(ref_inf2, ref_inf1)


if __name__ == '__main__':
    main()

"""Input pipeline kernels through the C ABI: yolo2_augment_images against the oracle's restatement of the
TF-1.0 image ops, yolo2_transform_labels bit-exact against the reference's goldens (tests/golden/labels.npz, generated
from the reference's transform_labels) and against the host port on ragged batches."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import yolo2_ref as R   # noqa: E402


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')
    from yolo_tf_amd import ops as o
    return o


def run_augment(ops, images, params, W, H):
    from yolo_tf_amd._lib import AugmentParams
    from yolo_tf_amd.utils import augment as A
    B = len(images)
    arr = (AugmentParams * B)()
    off = 0
    for d, im, p in zip(arr, images, params):
        d.src_offset, d.src_w, d.src_h = off, im.shape[1], im.shape[0]
        off += im.size
        d.crop_x, d.crop_y, d.crop_w, d.crop_h = p.get('crop') or (0, 0, im.shape[1], im.shape[0])
        fl = 0
        for key, bit in (('flip', A.FLIP), ('gray', A.GRAY)):
            if p.get(key):
                fl |= bit
        for key, bit in (('brightness', A.BRIGHTNESS), ('saturation', A.SATURATION), ('hue', A.HUE), ('contrast', A.CONTRAST)):
            if p.get(key) is not None:
                fl |= bit
                setattr(d, key, p[key])
        if p.get('noise_scale') is not None:
            fl |= A.NOISE
            d.noise_scale, d.noise_seed = p['noise_scale'], p.get('noise_seed', 7)
        d.flags = fl
    src = torch.from_numpy(np.concatenate([im.reshape(-1) for im in images])).cuda()
    pd = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    ws = torch.full((3 * B,), 9.0, dtype=torch.float64, device='cuda')          # dirty on purpose
    out = torch.zeros(B, H, W, 3, dtype=torch.float32, device='cuda')
    ops.augment_images(src, pd, ws, out, B, H, W, any(a.flags & A.CONTRAST for a in arr))
    torch.cuda.synchronize()
    return out.cpu().numpy()


CASES = [
    dict(),                                                            # plain bilinear resize
    dict(crop=(13, 9, 101, 77)),
    dict(flip=True),
    dict(brightness=-41.5),
    dict(saturation=1.37),
    dict(saturation=0.52, hue=-0.031),
    dict(hue=0.032),
    dict(contrast=0.61),
    dict(gray=True),
    dict(crop=(3, 5, 150, 99), flip=True, brightness=35.0, saturation=1.45, hue=0.02, contrast=1.42, gray=False),
    dict(crop=(0, 0, 64, 48), flip=True, brightness=60.0, saturation=0.7, hue=-0.01, contrast=0.55, gray=True),
]


@pytest.mark.parametrize('size', [(64, 48), (96, 128)])
def test_augment_images_matches_oracle(ops, size):
    W, H = size
    rng = np.random.RandomState(W)
    images = [rng.randint(0, 256, (120 + 3 * i, 160 + 5 * i, 3)).astype(np.uint8) for i in range(len(CASES))]
    images[4][:20, :20] = 128                                          # a flat grey patch: range 0 -> hue 0 / saturation 0 paths
    images[5][:10] = 0                                                 # black rows: V = 0
    got = run_augment(ops, images, CASES, W, H)
    for i, (im, p) in enumerate(zip(images, CASES)):
        ref = R.augment_image(im, p, W, H)
        err = np.abs(got[i] - ref).max()
        assert err <= 2e-3, 'case %d %s: max abs err %.3e on the 0..255 scale' % (i, p, err)      # f32 rounding through HSV
    assert got.min() >= 0 and got.max() <= 255


def test_augment_same_size_is_a_copy_and_crop_same_size_too(ops):
    rng = np.random.RandomState(0)
    im = rng.randint(0, 256, (48, 64, 3)).astype(np.uint8)
    big = rng.randint(0, 256, (100, 100, 3)).astype(np.uint8)
    got = run_augment(ops, [im, big], [dict(), dict(crop=(11, 7, 64, 48))], 64, 48)
    np.testing.assert_array_equal(got[0], im.astype(np.float32))       # resize_images' same-size shortcut: exact
    np.testing.assert_array_equal(got[1], big[7:55, 11:75].astype(np.float32))


def test_augment_noise_statistics_and_determinism(ops):
    im = np.full((64, 64, 3), 128, np.uint8)
    p = dict(noise_scale=10.0, noise_seed=1234567)
    a = run_augment(ops, [im], [p], 64, 64)[0] - 128.0
    b = run_augment(ops, [im], [p], 64, 64)[0] - 128.0
    c = run_augment(ops, [im], [dict(noise_scale=10.0, noise_seed=1234568)], 64, 64)[0] - 128.0
    np.testing.assert_array_equal(a, b)                                # counter-based: same seed, same noise
    assert np.abs(a - c).max() > 1.0
    z = a / 10.0
    assert np.abs(z).max() <= 2.0 + 1e-5                               # tf.truncated_normal: resampled beyond 2 sigma
    assert abs(z.mean()) < 0.03 and abs(z.std() - 0.8796) < 0.02       # std of N(0,1) truncated at 2
    assert abs(np.corrcoef(z[..., 0].ravel(), z[..., 1].ravel())[0, 1]) < 0.05


def _device_labels(ops, objs, classes, cw, ch):
    B = len(objs)
    cells = cw * ch
    cls = np.concatenate([np.asarray(c, np.int32).reshape(-1) for c, _ in objs] + [np.zeros(0, np.int32)])
    box = np.concatenate([np.asarray(b, np.float32).reshape(-1, 4) for _, b in objs] + [np.zeros((0, 4), np.float32)])
    first = np.cumsum([0] + [len(np.asarray(c).reshape(-1)) for c, _ in objs]).astype(np.int32)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    shapes = [(B, cells, 1), (B, cells, 1, classes), (B, cells, 1, 4), (B, cells, 1, 2), (B, cells, 1, 2), (B, cells, 1)]
    outs = [torch.full(s, 5.0, dtype=torch.float32, device='cuda') for s in shapes]     # dirty on purpose
    err = torch.zeros(1, dtype=torch.int32, device='cuda')
    ops.transform_labels(d(cls if len(cls) else np.zeros(1, np.int32)), d(box if len(box) else np.zeros((1, 4), np.float32)), d(first),
                         *outs, B, classes, cw, ch, err)
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in outs], int(err.item())


@pytest.mark.parametrize('case', ['voc13', 'coco13', 'rect', 'shared_cell'])
def test_transform_labels_bit_exact_vs_reference_golden(ops, case):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'labels.npz'))
    classes, cw, ch = [int(v) for v in g[case + '/dims']]
    outs, err = _device_labels(ops, [(g[case + '/class'], g[case + '/coord'])], classes, cw, ch)
    assert err == 0
    for o, key in zip(outs, ('mask', 'prob', 'coords', 'offset_xy_min', 'offset_xy_max', 'areas')):
        np.testing.assert_array_equal(o[0], g['%s/%s' % (case, key)], err_msg=key)


def test_transform_labels_ragged_batch_and_errors(ops):
    from yolo_tf_amd.utils import data
    rng = np.random.RandomState(4)
    objs = []
    for k in (3, 0, 7, 1, 12):                                         # an image without objects in the middle
        centre = rng.uniform(0.05, 0.95, (k, 2))
        size = rng.uniform(0.02, 0.5, (k, 2))
        box = np.clip(np.concatenate([centre - size / 2, centre + size / 2], 1), 0, 1).astype(np.float32)
        if k == 12:
            box[5] = box[4]                                            # two objects in one cell: the later one wins, both class bits stay
        objs.append((rng.randint(0, 20, k), box))
    outs, err = _device_labels(ops, objs, 20, 13, 13)
    assert err == 0
    for b, (c, box) in enumerate(objs):
        ref = data.transform_labels(c, box, 20, 13, 13)
        for o, r in zip(outs, ref):
            np.testing.assert_array_equal(o[b].reshape(r.shape), r)
    # the reference raises IndexError for a centre outside the grid and asserts on negative extents
    _, e1 = _device_labels(ops, [(np.array([1]), np.array([[0.9, 0.9, 1.2, 1.2]], np.float32))], 20, 13, 13)
    _, e2 = _device_labels(ops, [(np.array([25]), np.array([[0.1, 0.1, 0.2, 0.2]], np.float32))], 20, 13, 13)
    _, e3 = _device_labels(ops, [(np.array([1]), np.array([[0.5, 0.5, 0.4, 0.6]], np.float32))], 20, 13, 13)
    assert e1 & 1 and e2 & 1 and e3 & 2


def test_device_pipeline_feeds_a_training_step(ops, tmp_path):
    """DeviceInputPipeline -> TrainSession: augmented batch + labels written in place, loss finite and decreasing."""
    import configparser
    from test_network_gpu import make_builder
    from yolo_tf_amd.session import TrainSession
    from yolo_tf_amd.utils.augment import DeviceInputPipeline
    b, cfg = make_builder('tiny', 20, 96, True, str(tmp_path))
    sess = TrainSession(b, 4, dtype='f32', optimizer='adam', learning_rate=1e-3, seed=1)
    rng = np.random.RandomState(8)
    images = [rng.randint(0, 256, (70 + 10 * i, 90 + 7 * i, 3)).astype(np.uint8) for i in range(6)]
    objects = []
    for im in images:
        h, w = im.shape[:2]
        k = rng.randint(1, 4)
        x0, y0 = rng.uniform(0, 0.5 * w, k), rng.uniform(0, 0.5 * h, k)
        objects.append((rng.randint(0, 20, k), np.stack([x0, y0, x0 + rng.uniform(8, 0.4 * w, k), y0 + rng.uniform(8, 0.4 * h, k)], 1)))
    ini = configparser.ConfigParser()
    ini.read(os.path.join(ROOT, 'config.ini'))
    pipe = DeviceInputPipeline(images, objects, 4, 96, 96, 20, 3, 3, config=ini, seed=2)
    losses = []
    for _ in range(6):
        batch = pipe.next(sess)
        pipe.check()
        assert float(batch.min()) >= 0 and float(batch.max()) <= 255
        assert float(sess.labels[0].sum()) >= 1                        # at least one responsible cell per batch
        sess.step(batch)
        losses.append(sess.fetch()['total_loss'])
    assert np.all(np.isfinite(losses))


def test_reference_cache_to_training_step(ops, tmp_path):
    """The whole widened path: JPEGs + a TFRecord cache in the reference's schema (utils/tfrecord.py) -> HBM-resident dataset ->
    on-device augmentation and labels -> training steps."""
    Image = pytest.importorskip('PIL.Image')
    import configparser
    from test_network_gpu import make_builder
    from yolo_tf_amd.session import TrainSession
    from yolo_tf_amd.utils import tfrecord
    from yolo_tf_amd.utils.augment import DeviceInputPipeline
    rng = np.random.RandomState(11)
    samples = []
    for i in range(5):
        h, w = 60 + 9 * i, 80 + 11 * i
        p = str(tmp_path / ('%06d.jpg' % i))
        Image.fromarray(rng.randint(0, 256, (h, w, 3)).astype(np.uint8)).save(p, quality=90)
        k = rng.randint(1, 4)
        x0, y0 = rng.uniform(0, 0.5 * w, k), rng.uniform(0, 0.5 * h, k)
        samples.append((p, (h, w, 3), rng.randint(0, 20, k), np.stack([x0, y0, x0 + rng.uniform(8, 0.4 * w, k), y0 + rng.uniform(8, 0.4 * h, k)], 1)))
    cache = str(tmp_path / 'train.tfrecord')
    tfrecord.write_cache(cache, samples)
    images, objects = tfrecord.load_dataset([cache])
    assert len(images) == 5 and images[3].shape == (87, 113, 3)
    b, _ = make_builder('tiny', 20, 96, True, str(tmp_path / 'base'))
    sess = TrainSession(b, 4, dtype='bf16', optimizer='adam', learning_rate=1e-3, seed=1)
    ini = configparser.ConfigParser()
    ini.read(os.path.join(ROOT, 'config.ini'))
    pipe = DeviceInputPipeline(images, objects, 4, 96, 96, 20, 3, 3, config=ini, seed=5)
    for _ in range(4):
        sess.step(pipe.next(sess))
        pipe.check()
    assert np.isfinite(sess.fetch()['total_loss'])

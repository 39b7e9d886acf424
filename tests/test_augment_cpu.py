"""Input pipeline, CPU side: the oracle's restatement of the TF-1.0 image ops against known answers, and
the host logic of yolo_tf_amd.utils.augment (random draws, box transforms) against the oracle's restatement of the
reference's box arithmetic (utils/preprocess.py:28-51, utils/data/__init__.py:63-68,171)."""
import colorsys
import configparser
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import yolo2_ref as R   # noqa: E402


def test_resize_bilinear_known_answer_and_identity():
    x = np.array([[0, 1], [2, 3]], np.float32)[..., None].repeat(3, -1)
    y = R.resize_bilinear(x, 4, 4)[..., 0]
    # TF-1.0 ResizeBilinear, align_corners=False: in = out * 0.5, bottom/right taps clamp at the edge
    np.testing.assert_array_equal(y, np.array([[0, .5, 1, 1], [1, 1.5, 2, 2], [2, 2.5, 3, 3], [2, 2.5, 3, 3]], np.float32))
    z = np.random.RandomState(0).rand(7, 5, 3).astype(np.float32)
    assert R.resize_bilinear(z, 7, 5) is not None and np.array_equal(R.resize_bilinear(z, 7, 5), z)
    # downscale by an integer factor samples the top-left pixel of each block
    w = np.arange(8 * 6 * 3, dtype=np.float32).reshape(8, 6, 3)
    np.testing.assert_array_equal(R.resize_bilinear(w, 4, 3), w[::2, ::2])


def test_hsv_matches_colorsys_and_round_trips():
    rng = np.random.RandomState(1)
    x = rng.rand(500, 3).astype(np.float32)
    hsv = R.rgb_to_hsv(x)
    ref = np.array([colorsys.rgb_to_hsv(*p) for p in x.astype(np.float64)])
    assert np.abs(hsv - ref).max() < 1e-6
    back = R.hsv_to_rgb(hsv)
    assert np.abs(back - x).max() < 1e-6
    img = rng.randint(0, 256, (9, 11, 3)).astype(np.float32)        # the pipeline works on 0..255: V scales, H and S do not
    assert np.abs(R.hsv_to_rgb(R.rgb_to_hsv(img)) - img).max() < 2e-4
    gray = np.full((2, 2, 3), 77, np.float32)
    np.testing.assert_array_equal(R.rgb_to_hsv(gray)[..., :2], 0)   # range 0 -> hue 0, saturation 0


def test_colour_ops_known_answers():
    rng = np.random.RandomState(2)
    img = rng.randint(0, 256, (6, 7, 3)).astype(np.float32)
    np.testing.assert_allclose(R.adjust_contrast(img, 1.0), img, atol=2e-5)
    c = R.adjust_contrast(img, 0.5)
    np.testing.assert_allclose(c.mean((0, 1)), img.mean((0, 1)), rtol=1e-6)         # the per-channel mean is the fixed point
    np.testing.assert_allclose(c - c.mean((0, 1)), (img - img.mean((0, 1))) * 0.5, atol=1e-4)
    g = R.rgb_to_grayscale3(img)
    assert np.array_equal(g[..., 0], g[..., 1]) and np.array_equal(g[..., 1], g[..., 2])
    np.testing.assert_allclose(g[..., 0], img @ np.array([0.2989, 0.5870, 0.1140], np.float32), rtol=1e-6)
    s0 = R.adjust_saturation(img, 0.0)                                                # no saturation: every channel = V = max
    np.testing.assert_allclose(s0, img.max(-1, keepdims=True).repeat(3, -1), atol=1e-4)
    np.testing.assert_allclose(R.adjust_hue(img, 0.0), img, atol=3e-4)
    third = R.adjust_hue(np.array([[[200, 10, 10]]], np.float32), 1.0 / 3.0)          # red -> green
    np.testing.assert_allclose(third, [[[10, 200, 10]]], atol=1e-3)


def test_box_transforms_follow_the_reference_arithmetic():
    from yolo_tf_amd.utils import augment as A
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(ROOT, 'config.ini'))
    ac = A.AugmentConfig(cfg)
    assert ac.full_enable and ac.resized_enable and ac.random_crop == 0.9 and ac.grayscale_probability == 0.05   # reference config.ini:14-28

    class Fixed(object):   # replays a fixed stream of uniforms so both sides see the same draws
        def __init__(self, vals):
            self.vals = list(vals)

        def uniform(self, lo=0.0, hi=1.0, size=None):
            if size is None:
                return lo + (hi - lo) * self.vals.pop(0)
            return lo + (hi - lo) * np.array([self.vals.pop(0) for _ in range(size)])

        def randint(self, lo, hi):
            return 12345

    coord = np.array([[40, 30, 200, 180], [100, 90, 300, 220]], np.float32)
    u4 = [0.3, 0.6, 0.2, 0.9]
    # draws: crop taken (0.1 < 0.5), 4 crop uniforms, flip taken (0.2 < 0.5), the five colour branches and grayscale not taken
    p, norm = A.draw(ac, Fixed([0.1] + u4 + [0.2] + [0.9] * 6), (320, 240), coord, 416, 416)
    c2, crop, wh = R.random_crop_box(coord, (320, 240), np.array(u4, np.float32) * np.float32(0.9), 0.9)
    assert p['crop'] == crop and p['flags'] == A.FLIP
    expect = R.flip_coords(R.resize_coords(c2, wh, 416, 416), 416) / np.array([416, 416, 416, 416], np.float32)
    np.testing.assert_array_equal(norm, expect.astype(np.float32))
    assert np.all(norm[:, 2] >= norm[:, 0]) and np.all(norm >= 0) and np.all(norm <= 1)
    # nothing enabled -> whole image, boxes only rescaled
    p0, n0 = A.draw(A.AugmentConfig(None), Fixed([]), (320, 240), coord, 416, 416)
    assert p0['crop'] == (0, 0, 320, 240) and p0['flags'] == 0
    np.testing.assert_allclose(n0, coord / np.array([320, 240, 320, 240], np.float32), rtol=1e-6)


def test_augment_image_oracle_composition():
    rng = np.random.RandomState(3)
    src = rng.randint(0, 256, (50, 70, 3)).astype(np.uint8)
    out = R.augment_image(src, dict(crop=(5, 7, 40, 30), flip=True, brightness=20.0, contrast=1.3, gray=False), 32, 24)
    assert out.shape == (24, 32, 3) and out.dtype == np.float32 and out.min() >= 0 and out.max() <= 255
    step = R.resize_bilinear(src[7:37, 5:45].astype(np.float32), 24, 32)[:, ::-1] + np.float32(20)
    np.testing.assert_array_equal(out, np.clip(R.adjust_contrast(step, 1.3), 0, 255))

"""On-device input pipeline: the counterpart of the reference's load_image_labels
(utils/data/__init__.py:162-175) after JPEG decode.

The reference builds this stage out of TF ops fed by an input queue; here decoded uint8 images and their raw boxes sit
in HBM and two launches produce what the training step consumes: ``yolo2_augment_images`` (crop -> bilinear resize ->
flip -> brightness / saturation / hue / contrast / noise / grayscale -> clip) and ``yolo2_transform_labels`` (boxes ->
the six label tensors, written straight into the session's label buffers).  Every random decision the reference makes
with tf.random_uniform / tf.cond is drawn here on the host, per image, from a seeded NumPy generator, in the reference's
order; the config keys are the reference's (`[data_augmentation_full]`, `[data_augmentation_resized]`, config.ini:14-28).
The box arithmetic (random_crop utils/preprocess.py:28-42, resize factor utils/data/__init__.py:63-68, flip
utils/preprocess.py:45-51, normalisation utils/data/__init__.py:171) is host NumPy in float32: a few boxes per image.
"""
import numpy as np
import torch

from .. import ops
from .._lib import AugmentParams

FLIP, BRIGHTNESS, SATURATION, HUE, CONTRAST, NOISE, GRAY = 1, 2, 4, 8, 16, 32, 64


class AugmentConfig(object):
    """The two reference sections; missing sections / keys mean 'disabled' like `enable = 0`."""

    def __init__(self, config=None):
        def get(section, key, conv, default):
            if config is not None and config.has_section(section) and config.has_option(section, key):
                return conv(config.get(section, key))
            return default
        as_bool = lambda s: s.strip().lower() in ('1', 'true', 'yes', 'on')
        full, res = 'data_augmentation_full', 'data_augmentation_resized'
        self.full_enable = get(full, 'enable', as_bool, False)
        self.full_probability = get(full, 'enable_probability', float, 0.5)
        self.random_crop = get(full, 'random_crop', float, 0.0)
        self.resized_enable = get(res, 'enable', as_bool, False)
        self.probability = get(res, 'enable_probability', float, 0.5)
        self.flip = get(res, 'random_flip_horizontally', as_bool, False)
        self.brightness = get(res, 'random_brightness', as_bool, False)
        self.contrast = get(res, 'random_contrast', as_bool, False)
        self.saturation = get(res, 'random_saturation', as_bool, False)
        self.hue = get(res, 'random_hue', as_bool, False)
        self.noise = get(res, 'noise', as_bool, False)
        self.grayscale_probability = get(res, 'grayscale_probability', float, 0.0)


def draw(cfg, rng, src_wh, objects_coord, width, height):
    """One image's random decisions + the transformed boxes.

    src_wh = (w, h) of the decoded image, objects_coord [K,4] float pixels (xmin, ymin, xmax, ymax).
    Returns (dict for AugmentParams: crop, flags, scalars, seed; boxes normalised to [0,1] as float32 [K,4])."""
    f = np.float32
    coord = np.asarray(objects_coord, f).reshape(-1, 4)
    wh = np.array(src_wh, f)
    crop = (0, 0, int(src_wh[0]), int(src_wh[1]))
    if cfg.full_enable and cfg.random_crop > 0 and len(coord) and rng.uniform() < cfg.full_probability:
        # utils/preprocess.py:28-42
        xy_min, xy_max = coord[:, :2].min(0), coord[:, 2:].max(0)
        shrink = rng.uniform(0, cfg.random_crop, 4).astype(f) * np.concatenate([xy_min, wh - xy_max]).astype(f)
        _xy_min = shrink[:2]
        _wh = wh - shrink[2:] - _xy_min
        coord = coord - np.tile(_xy_min, 2)
        crop = (int(_xy_min[0]), int(_xy_min[1]), max(int(_wh[0]), 1), max(int(_wh[1]), 1))
        wh = _wh
    coord = coord * np.tile(np.array([width, height], f) / wh, 2)            # resize_image_objects
    p = dict(crop=crop, flags=0, brightness=0.0, saturation=1.0, hue=0.0, contrast=1.0, noise_scale=0.0, noise_seed=0)
    if cfg.resized_enable:
        if cfg.flip and rng.uniform() < 0.5:                                 # random_flip_horizontally, probability 0.5
            p['flags'] |= FLIP
            w = f(width)
            coord = np.stack([w - coord[:, 2], coord[:, 1], w - coord[:, 0], coord[:, 3]], 1)
        if cfg.brightness and rng.uniform() < cfg.probability:
            p['flags'] |= BRIGHTNESS
            p['brightness'] = float(rng.uniform(-63, 63))
        if cfg.saturation and rng.uniform() < cfg.probability:
            p['flags'] |= SATURATION
            p['saturation'] = float(rng.uniform(0.5, 1.5))
        if cfg.hue and rng.uniform() < cfg.probability:
            p['flags'] |= HUE
            p['hue'] = float(rng.uniform(-0.032, 0.032))
        if cfg.contrast and rng.uniform() < cfg.probability:
            p['flags'] |= CONTRAST
            p['contrast'] = float(rng.uniform(0.5, 1.5))
        if cfg.noise and rng.uniform() < cfg.probability:
            p['flags'] |= NOISE
            p['noise_scale'] = float(rng.uniform(5, 15))
            p['noise_seed'] = int(rng.randint(0, 2 ** 31 - 1)) * 4294967291 + int(rng.randint(0, 2 ** 31 - 1))
        if cfg.grayscale_probability > 0 and rng.uniform() < cfg.grayscale_probability:
            p['flags'] |= GRAY
    coord = (coord / np.array([width, height, width, height], f)).astype(f)  # utils/data/__init__.py:171
    return p, coord


class DeviceInputPipeline(object):
    """Keeps a dataset of decoded images in HBM and assembles training batches there.

    images: list of uint8 [h, w, 3] arrays (any sizes); objects: list of (classes int [K], coords float [K,4] pixels).
    ``next(session)`` fills the session's label buffers in place and returns the f32 [B,H,W,3] image batch (0..255) that
    ``TrainSession.step`` standardises."""

    def __init__(self, images, objects, batch, width, height, classes, cell_width, cell_height, config=None, seed=0, rank=0, world=1):
        assert len(images) == len(objects) and len(images) > 0
        self.cfg = config if isinstance(config, AugmentConfig) else AugmentConfig(config)
        self.B, self.W, self.H, self.classes = batch, width, height, classes
        self.cell_width, self.cell_height = cell_width, cell_height
        self.rng = np.random.RandomState(seed)
        self.rank, self.world = rank, world
        self.sizes = [(im.shape[1], im.shape[0]) for im in images]
        offsets = np.cumsum([0] + [im.shape[0] * im.shape[1] * 3 for im in images])
        self.offsets = offsets[:-1]
        packed = np.concatenate([np.ascontiguousarray(im, np.uint8).reshape(-1) for im in images])
        self.src = torch.from_numpy(packed).cuda()
        self.objects = [(np.asarray(c, np.int32).reshape(-1), np.asarray(b, np.float32).reshape(-1, 4)) for c, b in objects]
        dev = self.src.device
        self.out = torch.zeros(batch, height, width, 3, dtype=torch.float32, device=dev)
        self.ws = torch.zeros(3 * batch, dtype=torch.float64, device=dev)
        self.err = torch.zeros(1, dtype=torch.int32, device=dev)

    def sample(self):
        idx = self.rng.randint(0, len(self.sizes), self.B * self.world)
        return idx[self.rank::self.world]

    def assemble(self, indices):
        """Host part: random draws + box transforms.  Returns (AugmentParams array, classes, coords, first_object)."""
        arr = (AugmentParams * self.B)()
        cls, box, first = [], [], [0]
        for d, i in zip(arr, indices):
            c, b = self.objects[i]
            p, nb = draw(self.cfg, self.rng, self.sizes[i], b, self.W, self.H)
            d.src_offset = int(self.offsets[i])
            d.src_w, d.src_h = self.sizes[i]
            d.crop_x, d.crop_y, d.crop_w, d.crop_h = p['crop']
            d.flags = p['flags']
            d.brightness, d.saturation, d.hue, d.contrast = p['brightness'], p['saturation'], p['hue'], p['contrast']
            d.noise_scale, d.noise_seed = p['noise_scale'], p['noise_seed'] & (2 ** 64 - 1)
            cls.append(c)
            box.append(nb)
            first.append(first[-1] + len(c))
        return arr, np.concatenate(cls).astype(np.int32), np.concatenate(box).astype(np.float32).reshape(-1, 4), np.asarray(first, np.int32)

    def launch(self, arr, cls, box, first, labels):
        """Device part: both kernels; ``labels`` = the six device tensors of the loss (TrainSession.labels)."""
        params = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
        any_contrast = any(a.flags & CONTRAST for a in arr)
        ops.augment_images(self.src, params, self.ws, self.out, self.B, self.H, self.W, any_contrast)
        n = max(len(cls), 1)
        cls_d = torch.from_numpy(np.resize(cls, n) if len(cls) else np.zeros(1, np.int32)).cuda()
        box_d = torch.from_numpy(box if len(box) else np.zeros((1, 4), np.float32)).cuda()
        first_d = torch.from_numpy(first).cuda()
        # the flag is sticky (the kernel ORs into it): check() reports a bad object of ANY batch since the last check
        ops.transform_labels(cls_d, box_d, first_d, *labels, self.B, self.classes, self.cell_width, self.cell_height, self.err)
        self._keep = (params, cls_d, box_d, first_d)      # alive until the next launch (asynchronous kernels)
        return self.out

    def next(self, session):
        arr, cls, box, first = self.assemble(self.sample())
        return self.launch(arr, cls, box, first, session.labels)

    def check(self):
        """Raises like the reference would (IndexError / AssertionError in transform_labels); synchronises."""
        e = int(self.err.item())
        self.err.zero_()
        if e & 1:
            raise IndexError('transform_labels: object outside the grid or class id out of range')
        if e & 2:
            raise AssertionError('transform_labels: negative box extent')


"""Config / path / plugin helpers with the reference's names and semantics (utils/__init__.py:28-73)."""
import importlib
import os

import numpy as np


def _expand(path):
    return os.path.expanduser(os.path.expandvars(path))


def get_cachedir(config):
    """<basedir>/cache/<basename of [cache] names>  (reference :28-31)."""
    return os.path.join(_expand(config.get('config', 'basedir')), 'cache', os.path.basename(config.get('cache', 'names')))


def get_logdir(config):
    """<basedir>/<model>/<inference>/<basename(names)>  (reference :34-39)."""
    model = config.get('config', 'model')
    return os.path.join(_expand(config.get('config', 'basedir')), model, config.get(model, 'inference'),
                        os.path.basename(config.get('cache', 'names')))


def _inference_module(config):
    return importlib.import_module('yolo_tf_amd.model.%s.inference' % config.get('config', 'model'))


def get_inference(config):
    return getattr(_inference_module(config), config.get(config.get('config', 'model'), 'inference'))


def get_downsampling(config):
    name = config.get(config.get('config', 'model'), 'inference')
    return getattr(_inference_module(config), name.upper() + '_DOWNSAMPLING')


def calc_cell_width_height(config, width, height):
    dw, dh = get_downsampling(config)
    assert width % dw == 0
    assert height % dh == 0
    return width // dw, height // dh


def load_config(config, paths):
    """INI overlays, later files win (reference :69-73, README.md:21)."""
    for path in paths:
        path = _expand(path)
        assert os.path.exists(path)
        config.read(path)


def read_anchors(path):
    """Anchor TSV with a `w h` header row, values in cell units (config/yolo2/anchors/*.tsv)."""
    return np.loadtxt(_expand(path), delimiter='\t', skiprows=1, dtype=np.float32).reshape(-1, 2)


def ensure_names(config):
    """Makes <cachedir>/names exist by copying the [cache] names list, which is the one thing the
    reference's cache.py (:28-48, dataset ETL, out of scope here) does that the model builder needs."""
    import shutil
    cachedir = get_cachedir(config)
    dst = os.path.join(cachedir, 'names')
    if not os.path.exists(dst):
        os.makedirs(cachedir, exist_ok=True)
        shutil.copyfile(_expand(config.get('cache', 'names')), dst)
    return dst


def make_config(paths, basedir=None):
    """ConfigParser from overlay files (+ optional basedir override) -- convenience for scripts/tests."""
    import configparser
    config = configparser.ConfigParser()
    load_config(config, paths)
    if basedir is not None:
        config.set('config', 'basedir', basedir)
    return config

"""The reference's two entry points as a user runs them (train.py:148-170, detect.py:108-124): train a few steps from the config
overlays, find the checkpoint + event file in the logdir the reference would use, then detect on an image file from that logdir.
Both model families, both checkpoint containers."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, timeout=600):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable] + cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, 'FAILED: %s\n--- stdout\n%s\n--- stderr\n%s' % (' '.join(cmd), r.stdout[-3000:], r.stderr[-3000:])
    return r.stdout + r.stderr


@pytest.mark.parametrize('model_ini,fmt,logsub', [('config/yolo2/darknet-20.ini', 'tf', ('yolo2', 'darknet')), ('config/yolo/tiny-20.ini', 'npz', ('yolo', 'tiny'))])
def test_train_then_detect_from_the_logdir(tmp_path, model_ini, fmt, logsub):
    from PIL import Image
    overlay = tmp_path / 'local.ini'
    overlay.write_text('[config]\nbasedir = %s\n' % tmp_path)
    cfg = ['-c', 'config.ini', model_ini, str(overlay)]
    out = run(['train.py'] + cfg + ['--data', 'synthetic', '-b', '2', '-s', '3', '-d', '--seed', '1', '--ckpt_format', fmt, '-n', 'run0', '--level', 'info'])
    logdir = os.path.join(str(tmp_path), logsub[0], logsub[1], '20')       # <basedir>/<model>/<inference>/<basename of [cache] names> (utils/__init__.py:34-39)
    assert os.path.isdir(logdir), (os.listdir(str(tmp_path)), out[-1500:])
    files = os.listdir(logdir)
    if fmt == 'tf':      # what tf.train.Saver leaves behind (train.py:141-145 / slim.learning.train)
        assert 'checkpoint' in files and any(f.startswith('model.ckpt-3') and f.endswith('.index') for f in files), files
        assert any(f.startswith('model.ckpt-3.data-00000-of-00001') for f in files), files
    else:
        assert any('3' in f and f.endswith('.npz') for f in files), files
    assert glob.glob(os.path.join(logdir, 'run0', 'events.out.tfevents.*')), os.listdir(logdir)
    img = tmp_path / 'img.jpg'
    Image.fromarray(np.random.RandomState(0).randint(0, 255, (375, 500, 3), dtype=np.uint8)).save(str(img))
    out = run(['detect.py', str(img)] + cfg + ['-t', '0.000001', '--level', 'info'])
    assert 'objects detected' in out and 'global_step=3' in out, out[-2000:]

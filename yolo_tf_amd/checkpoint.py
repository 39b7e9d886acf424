"""Checkpoint / resume for the training CLI.  Keeps the reference's directory layout and variable
names (logdir = <basedir>/<model>/<inference>/<names>, utils/__init__.py:34-39; variable scopes as in
parse_darknet_yolo2.py:71) in a self-describing ``model.ckpt-<step>.npz`` container; TF-checkpoint
binary compatibility is SURVEY 8(f) rank 4 (not built)."""
import glob
import os
import re

import numpy as np
import torch


def latest_checkpoint(logdir):
    best, best_step = None, -1
    for path in glob.glob(os.path.join(logdir, 'model.ckpt-*.npz')):
        m = re.search(r'model\.ckpt-(\d+)\.npz$', path)
        if m and int(m.group(1)) > best_step:
            best, best_step = path, int(m.group(1))
    return best


def save(logdir, session):
    os.makedirs(logdir, exist_ok=True)
    e = session.engine
    data = {'var/' + k: v for k, v in e.get_variables().items()}
    data['global_step'] = np.int64(session.global_step)
    data['optimizer'] = np.array(session.optimizer.name)
    for i, s in enumerate(session.optimizer.slots):
        data['slot/%d' % i] = s.cpu().numpy()
    data['param_layout'] = np.array(sorted((o, n, k) for k, (o, n) in e.param_offsets.items()), dtype=object).astype(str)
    path = os.path.join(logdir, 'model.ckpt-%d.npz' % session.global_step)
    tmp = path + '.tmp.npz'
    np.savez(tmp, **data)
    os.replace(tmp, path)
    return path


def restore(path, session=None, engine=None, exclude=None, variables_only=False):
    """Restores variables (all but those whose name starts with a scope in ``exclude`` -- the
    reference's ``-t ckpt -e scope...`` transfer, train.py:114,130-136) and, for a full resume,
    optimizer slots + global_step."""
    z = np.load(path, allow_pickle=False)
    engine = engine if engine is not None else session.engine
    values = {}
    for k in z.files:
        if k.startswith('var/'):
            name = k[4:]
            if exclude and any(name.startswith(s) for s in exclude):
                continue
            values[name] = z[k]
    engine.set_variables(values, strict=False)
    step = int(z['global_step'])
    if session is not None and not variables_only:
        if str(z['optimizer']) == session.optimizer.name:
            for i, s in enumerate(session.optimizer.slots):
                s.copy_(torch.from_numpy(z['slot/%d' % i]))
        session.global_step = step
    return step

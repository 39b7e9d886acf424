"""Checkpoint / resume for the training CLI.  Keeps the reference's directory layout and variable
names (logdir = <basedir>/<model>/<inference>/<names>, utils/__init__.py:34-39; variable scopes as in
parse_darknet_yolo2.py:71) in a self-describing ``model.ckpt-<step>.npz`` container; TF-checkpoint
binary compatibility lives in tf_checkpoint.py."""
import glob
import logging
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


def _layout(engine):
    return np.array(sorted((o, n, k) for k, (o, n) in engine.param_offsets.items()), dtype=object).astype(str)


def save(logdir, session, keep=5):
    """Writes model.ckpt-<step>.npz atomically and keeps the ``keep`` most recent files ([TF-sem] tf.train.Saver
    max_to_keep=5, which slim.learning.train's default saver uses; each file is ~0.8 GB with Adam slots)."""
    if not getattr(session, 'optimizer_state_complete', True):
        raise RuntimeError('optimizer sharding: this rank holds the optimizer slots of its own shards only; call '
                           'session.gather_optimizer_state() on EVERY rank before saving (train.py does)')
    os.makedirs(logdir, exist_ok=True)
    e = session.engine
    data = {'var/' + k: v for k, v in e.get_variables().items()}
    data['global_step'] = np.int64(session.global_step)
    data['optimizer'] = np.array(session.optimizer.name)
    for i, s in enumerate(session.optimizer.slots):
        data['slot/%d' % i] = s.cpu().numpy()
    data['param_layout'] = _layout(e)
    path = os.path.join(logdir, 'model.ckpt-%d.npz' % session.global_step)
    tmp = path + '.tmp.npz'
    np.savez(tmp, **data)
    os.replace(tmp, path)
    if keep and keep > 0:
        found = []
        for p in glob.glob(os.path.join(logdir, 'model.ckpt-*.npz')):
            m = re.search(r'model\.ckpt-(\d+)\.npz$', p)
            if m:
                found.append((int(m.group(1)), p))
        for _, p in sorted(found)[:-keep]:
            if p != path:
                os.remove(p)
    return path


def restore(path, session=None, engine=None, exclude=None, variables_only=False):
    """Restores variables (all but those whose name starts with a scope in ``exclude`` -- the
    reference's ``-t ckpt -e scope...`` transfer, train.py:114,130-136) and, for a full resume,
    optimizer slots + global_step.  A transfer (``variables_only``) still restores ``global_step`` unless ``exclude``
    names it: the reference's slim.get_variables_to_restore(exclude=...) includes the global step, so the
    exponential_decay schedule continues from the donor's step.  Optimizer slots are flat arenas in the engine's
    parameter layout: they are copied only when the saved layout equals the current one (a different model, class count or
    variable order would silently misalign the moments)."""
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
        same_layout = 'param_layout' in z.files and np.array_equal(z['param_layout'], _layout(engine))
        if str(z['optimizer']) != session.optimizer.name:
            logging.warning('%s was written by optimizer %s: %s slots start fresh', path, z['optimizer'], session.optimizer.name)
        elif not same_layout:
            # Same variables at other offsets (the arena's alignment changed between builds): move the moments variable by variable.
            # Anything else (another model, class count, variable set) would misalign them: start fresh.
            saved = {str(k): (int(o), int(n)) for o, n, k in z['param_layout']} if 'param_layout' in z.files else {}
            cur = engine.param_offsets
            if saved and set(saved) == set(cur) and all(saved[k][1] == cur[k][1] for k in cur):
                for i, s in enumerate(session.optimizer.slots):
                    old = z['slot/%d' % i]
                    new = np.zeros(s.numel(), old.dtype)
                    for k, (o, n) in cur.items():
                        new[o:o + n] = old[saved[k][0]:saved[k][0] + n]
                    s.copy_(torch.from_numpy(new))
                logging.info('%s: optimizer slots remapped to the current arena offsets', path)
            else:
                logging.warning('%s has a different parameter layout: optimizer slots start fresh', path)
        else:
            for i, s in enumerate(session.optimizer.slots):
                s.copy_(torch.from_numpy(z['slot/%d' % i]))
        session.global_step = step
    elif session is not None and not (exclude and any('global_step'.startswith(s) for s in exclude)):
        session.global_step = step
    return step

"""Optimizers and learning-rate schedule with the reference's configuration surface
(train.py:70-80 get_optimizer, :116-124 exponential_decay; config.ini [optimizer_*],
[exponential_decay]).  Each optimizer is ONE kernel launch over the flat parameter arena
(csrc/elementwise.hip), with TF-1.0 Apply* update rules."""
import configparser
import math

import torch

from . import ops


def learning_rate_fn(config, base_lr):
    """lr * decay_rate ** floor(step / decay_steps) when [exponential_decay] is present, else constant
    (reference train.py:116-124)."""
    try:
        steps = config.getint('exponential_decay', 'decay_steps')
        rate = config.getfloat('exponential_decay', 'decay_rate')
        staircase = config.getboolean('exponential_decay', 'staircase')
    except (configparser.NoSectionError, configparser.NoOptionError, AttributeError):
        return lambda step: base_lr

    def fn(step):
        p = step / steps
        if staircase:
            p = math.floor(p)
        return base_lr * rate ** p
    return fn


DEFAULTS = {
    'adam': {'beta1': 0.9, 'beta2': 0.999, 'epsilon': 1e-8},
    'adadelta': {'rho': 0.95, 'epsilon': 1e-8},
    'adagrad': {'initial_accumulator_value': 0.1},
    'momentum': {'momentum': 0.9},
    'rmsprop': {'decay': 0.9, 'momentum': 0.0, 'epsilon': 1e-10},
    'ftrl': {'learning_rate_power': -0.5, 'initial_accumulator_value': 0.1, 'l1_regularization_strength': 0.0, 'l2_regularization_strength': 0.0},
    'gd': {},
}


class Optimizer(object):
    def __init__(self, name, config, n, device):
        if name not in DEFAULTS:
            raise KeyError('optimizer %r is not available on this path (adam, adadelta, adagrad, momentum, rmsprop, ftrl, gd)' % name)
        self.name = name
        self.hp = dict(DEFAULTS[name])
        section = 'optimizer_' + name
        if config is not None and config.has_section(section):
            for key in self.hp:
                if config.has_option(section, key):
                    self.hp[key] = config.getfloat(section, key)
        self.n = n
        from .engine import staggered          # (arena placement: see engine.staggered)
        z = lambda fill=0.0: staggered(n, torch.float32, device, fill)
        self.slots = {
            'adam': lambda: [z(), z()], 'adadelta': lambda: [z(), z()],
            'adagrad': lambda: [z(self.hp.get('initial_accumulator_value', 0.1))], 'momentum': lambda: [z()],
            'rmsprop': lambda: [z(1.0), z()],          # [TF-sem] RMSProp's `rms` slot starts at ones
            'ftrl': lambda: [z(self.hp['initial_accumulator_value']), z()],     # accum, linear
            'gd': lambda: [],
        }[name]()

    def apply_fused_with_filter_prep(self, engine, lr, t, gscale=1.0):
        """Adam over the whole arena and the operand layouts of the updated filters in one launch (engine.adam_update_and_prepare);
        bit-identical to ``apply`` followed by the engine's filter preparation."""
        assert self.name == 'adam'
        hp = self.hp
        alpha = lr * math.sqrt(1.0 - hp['beta2'] ** t) / (1.0 - hp['beta1'] ** t)
        engine.adam_update_and_prepare(self.slots[0], self.slots[1], alpha, hp['beta1'], hp['beta2'], hp['epsilon'], gscale)

    def apply(self, params, grads, lr, t, gscale=1.0, lo=0, hi=None):
        """t = 1-based update count (Adam bias correction).  [lo, hi) restricts the update to a slice of the flat arenas
        (data parallel: one call per all-reduce bucket as the buckets arrive)."""
        hp = self.hp
        hi = self.n if hi is None else hi
        n = hi - lo
        if lo != 0 or hi != self.n:
            params, grads = params[lo:hi], grads[lo:hi]
        s = [slot[lo:hi] for slot in self.slots] if (lo != 0 or hi != self.n) else self.slots
        if self.name == 'adam':
            alpha = lr * math.sqrt(1.0 - hp['beta2'] ** t) / (1.0 - hp['beta1'] ** t)
            ops.adam(params, grads, s[0], s[1], n, alpha, hp['beta1'], hp['beta2'], hp['epsilon'], gscale)
        elif self.name == 'momentum':
            ops.momentum(params, grads, s[0], n, lr, hp['momentum'], gscale)
        elif self.name == 'gd':
            ops.sgd(params, grads, n, lr, gscale)
        elif self.name == 'rmsprop':
            ops.rmsprop(params, grads, s[0], s[1], n, lr, hp['decay'], hp['momentum'], hp['epsilon'], gscale)
        elif self.name == 'adagrad':
            ops.adagrad(params, grads, s[0], n, lr, gscale)
        elif self.name == 'ftrl':
            ops.ftrl(params, grads, s[0], s[1], n, lr, hp['learning_rate_power'], hp['l1_regularization_strength'], hp['l2_regularization_strength'], gscale)
        elif self.name == 'adadelta':
            ops.adadelta(params, grads, s[0], s[1], n, lr, hp['rho'], hp['epsilon'], gscale)

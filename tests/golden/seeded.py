"""Seeded variable values shared by the fixture generator (make_golden.py) and the tests that replay the fixtures:
Darknet-19's 67 M parameters are not committed -- each variable is regenerated from its NAME, so the values do not depend
on creation order or on which side (reference-under-shim, oracle, engine) asks for them."""
import zlib

import numpy as np


def value(name, shape, kind, seed=0):
    """kind: weights | gamma | beta | moving_mean | moving_variance | biases."""
    rng = np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0xffffffff)
    shape = tuple(int(s) for s in shape)
    if kind == 'weights':
        fan = float(np.prod(shape[:-1]))                # k*k*cin (convolution) or in_features (fully connected)
        return (rng.standard_normal(shape) * np.sqrt(2.0 / fan)).astype(np.float32)
    if kind in ('gamma', 'moving_variance'):
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    return (rng.standard_normal(shape) * 0.1).astype(np.float32)


def kind_of(name):
    leaf = name.rsplit('/', 1)[-1]
    return leaf if leaf in ('weights', 'gamma', 'beta', 'moving_mean', 'moving_variance', 'biases') else 'biases'

"""Darknet .weights import/export: channel permutation pinned to the reference's own
transpose_weights / transpose_biases (tests/golden/weights.npz), file layout restated from
parse_darknet_yolo2.py:79-101 by an independent writer in this test."""
import os
import struct

import numpy as np
import pytest

from yolo_tf_amd import darknet_weights as D
from yolo_tf_amd import graph as G
from yolo_tf_amd.model.yolo2 import inference


def test_final_layer_permutation_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'weights.npz'))
    for name in ('voc', 'coco', 'a3c4'):
        a = int(g[name + '/anchors'])
        assert np.array_equal(D.transpose_weights(g[name + '/w_in'], a), g[name + '/w_out'])
        assert np.array_equal(D.transpose_biases(g[name + '/b_in'], a), g[name + '/b_out'])
        assert np.array_equal(D.transpose_weights(g[name + '/w_out'], a, inverse=True), g[name + '/w_in'])
        assert np.array_equal(D.transpose_biases(g[name + '/b_out'], a, inverse=True), g[name + '/b_in'])


@pytest.mark.parametrize('plugin', ['tiny', '_tiny', 'darknet'])
def test_weights_file_round_trip(tmp_path, plugin):
    classes, anchors = 20, 5
    g = G.Graph()
    getattr(inference, plugin)(G.placeholder(g, 'image', 416, 416), classes, anchors)
    convs = [op for op in g.ops if op['kind'] == 'conv']
    rng = np.random.RandomState(0)
    # independent writer: header, then per layer [biases|beta, gamma, mean, var] + OIHW weights, Darknet head order
    path = str(tmp_path / 'net.weights')
    expect = {}
    with open(path, 'wb') as f:
        f.write(struct.pack('4i', 0, 1, 0, 12345))
        for i, op in enumerate(convs):
            k, cin, cout = op['ksize'], op['cin'], op['cout']
            scope = op['name']
            if op['bn']:
                first = 'BatchNorm/beta' if scope + '/BatchNorm/beta' in g.variables else 'biases'
                for suffix in (first, 'BatchNorm/gamma', 'BatchNorm/moving_mean', 'BatchNorm/moving_variance'):
                    v = rng.randn(cout).astype(np.float32)
                    f.write(v.tobytes())
                    expect[scope + '/' + suffix] = v
                w = rng.randn(cout, cin, k, k).astype(np.float32)
                f.write(w.tobytes())
                expect[scope + '/weights'] = w.transpose(2, 3, 1, 0)
            else:
                b = rng.randn(cout).astype(np.float32)
                w = rng.randn(cout, cin, k, k).astype(np.float32)
                f.write(b.tobytes())
                f.write(w.tobytes())
                per = cout // anchors
                perm = np.concatenate([np.array([4, 0, 1, 2, 3] + list(range(5, per))) + a * per for a in range(anchors)])
                expect[scope + '/biases'] = b[perm]
                expect[scope + '/weights'] = w.transpose(2, 3, 1, 0)[..., perm]
        f.write(b'\0' * 8)                                   # trailing bytes are reported, not fatal
    header, values = D.load(path, g, anchors)
    assert header['seen'] == 12345 and header['remaining'] == 8
    assert set(values) == set(g.variables)
    for k, v in expect.items():
        assert np.array_equal(values[k], v), k
    out = str(tmp_path / 'out.weights')
    D.save(out, g, values, anchors, header=(0, 1, 0, 12345))
    assert open(out, 'rb').read() == open(path, 'rb').read()[:-8]
    with open(path, 'r+b') as f:
        f.truncate(1000)
    with pytest.raises(ValueError):
        D.load(path, g, anchors)

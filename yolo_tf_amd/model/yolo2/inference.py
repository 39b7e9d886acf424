"""YOLOv2 inference plugins: ``fn(net, classes, num_anchors, training=False) -> (scope, net)`` plus a
``<NAME>_DOWNSAMPLING`` constant, selected by ``[yolo2] inference`` (reference
model/yolo2/__init__.py:107, utils/__init__.py:47-49).

Topologies restate reference model/yolo2/inference.py:25-50 (tiny) and :61-120 (darknet = Darknet-19
backbone + 3x3 head + passthrough/reorg/concat, the early yolo.cfg without the 1x1 squeeze on the
passthrough, hence the 3072-channel conv20).  Every ``conv<i>`` is conv(no bias) -> batch_norm(eps 1e-5)
-> leaky_relu(0.1), executed as one fused op group on the MI355X; the last ``conv`` is a biased 1x1
without normalisation or activation.  Variable scopes are ``yolo2_<fn>/conv<i>/...`` as in the
reference (README.md:56)."""
from ... import graph as G
from .function import reorg

# (kernel, channel multiplier relative to the running width) blocks of Darknet-19 after the two stem convs
_BOTTLENECK3 = ((3, 1), (1, 0.5), (3, 1))
_BOTTLENECK5 = ((3, 1), (1, 0.5), (3, 1), (1, 0.5), (3, 1))


class _Scope(object):
    """Numbers the conv / pool scopes the way the reference's running `index` does."""

    def __init__(self, scope, center, init=G.xavier_uniform):
        self.scope, self.center, self.init, self.index = scope, center, init, 0

    def conv(self, net, channels, ksize=3):
        net = G.conv2d(net, channels, ksize, scope='%s/conv%d' % (self.scope, self.index), center=self.center,
                       weights_initializer=self.init)
        return net

    def pool(self, net, stride=2):
        return G.max_pool2d(net, stride=stride, scope='%s/max_pool%d' % (self.scope, self.index))

    def head(self, net, classes, num_anchors):
        return G.conv2d(net, num_anchors * (5 + classes), 1, scope='%s/conv' % self.scope, batch_norm=False, activation=False)


def tiny(net, classes, num_anchors, training=False, center=True):
    scope = 'yolo2_tiny'
    s = _Scope(scope, center, init=G.truncated_normal(0.1))
    channels = 16
    for _ in range(5):                       # conv0-4, each followed by a stride-2 pool
        net = s.pool(s.conv(net, channels))
        s.index += 1
        channels *= 2
    net = s.pool(s.conv(net, channels), stride=1)   # conv5 + the stride-1 SAME pool (13 -> 13)
    s.index += 1
    channels *= 2
    net = s.conv(net, channels)                     # conv6
    s.index += 1
    net = s.conv(net, channels)                     # conv7
    net = s.head(net, classes, num_anchors)
    return scope, net


TINY_DOWNSAMPLING = (2 ** 5, 2 ** 5)


def _tiny(net, classes, num_anchors, training=False):
    """Darknet weight layout variant: biases instead of beta (reference :55-58)."""
    scope, net = tiny(net, classes, num_anchors, training, center=False)
    return scope, net


_TINY_DOWNSAMPLING = (2 ** 5, 2 ** 5)


def darknet(net, classes, num_anchors, training=False, center=True):
    scope = 'yolo2_darknet'
    s = _Scope(scope, center)
    channels = 32
    for _ in range(2):                               # conv0, conv1 (+pool)
        net = s.pool(s.conv(net, channels))
        s.index += 1
        channels *= 2
    for _ in range(2):                               # conv2-4, conv5-7 (+pool)
        for i, (k, mult) in enumerate(_BOTTLENECK3):
            net = s.conv(net, channels * mult, k)
            if i + 1 < len(_BOTTLENECK3):
                s.index += 1
        net = s.pool(net)
        s.index += 1
        channels *= 2
    for i, (k, mult) in enumerate(_BOTTLENECK5):     # conv8-12
        net = s.conv(net, channels * mult, k)
        if i + 1 < len(_BOTTLENECK5):
            s.index += 1
    passthrough = G.identity(net, scope + '/passthrough')
    net = s.pool(net)
    s.index += 1
    channels *= 2                                    # 1024: downsampling finished (13x13 at 416)
    for k, mult in _BOTTLENECK5 + ((3, 1), (3, 1)):  # conv13-19
        net = s.conv(net, channels * mult, k)
        s.index += 1
    moved = reorg(passthrough, name=scope + '/reorg')
    net = G.concat([moved, net], 3, name='%s/concat%d' % (scope, s.index))   # reorg output first
    net = s.conv(net, channels)                      # conv20 on 3072 channels
    net = s.head(net, classes, num_anchors)
    return scope, net


DARKNET_DOWNSAMPLING = (2 ** 5, 2 ** 5)


def _darknet(net, classes, num_anchors, training=False):
    scope, net = darknet(net, classes, num_anchors, training, center=False)
    return scope, net


_DARKNET_DOWNSAMPLING = (2 ** 5, 2 ** 5)

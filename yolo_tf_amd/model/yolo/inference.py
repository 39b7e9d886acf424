"""YOLO (v1) inference plugins: ``fn(net, classes, boxes_per_cell, training=False) -> (scope, net)`` and ``<NAME>_DOWNSAMPLING``
(reference model/yolo/inference.py:24-66; selected by ``[yolo] inference``).

``tiny``: nine 3x3 convolutions WITH biases and leaky_relu (no batch norm: slim.layers.conv2d without a normalizer_fn), 2x2 max
pools after the first six (448 -> 7), flatten in (h, w, c) order, fully connected 256 and 4096 (leaky_relu, dropout 0.5 in
training, l2 regulariser 0.001 on the weights) and a linear, un-regularised fully connected output of cells * (classes + boxes_per_cell * 5)
(pinned by tests/golden/topology.json, recorded from the reference's own function).
Variable scopes ``yolo_tiny/conv<i>/{weights,biases}``, ``yolo_tiny/fc<i>/...``, ``yolo_tiny/fc/...`` as in the reference."""
from ... import graph as G


def tiny(net, classes, boxes_per_cell, training=False):
    scope = 'yolo_tiny'
    index = 0
    for channels, pooled in ((16, True), (32, True), (64, True), (128, True), (256, True), (512, True), (512, False), (1024, False), (256, False)):
        net = G.conv2d(net, channels, 3, scope='%s/conv%d' % (scope, index), batch_norm=False, activation=True)
        if pooled:
            net = G.max_pool2d(net, scope='%s/max_pool%d' % (scope, index))
        index += 1
    cell_height, cell_width = net.h, net.w          # `<scope>/conv`: the grid the Model reads back (model/yolo/__init__.py:39)
    net.graph.cells_hw = (cell_height, cell_width)
    net = G.flatten(net, scope='%s/flatten' % scope)
    index = 0
    for units in (256, 4096):
        net = G.fully_connected(net, units, scope='%s/fc%d' % (scope, index), weights_regularizer=0.001)
        net = G.dropout(net, 0.5, is_training=training, scope='%s/dropout%d' % (scope, index))
        index += 1
    # the output layer sits OUTSIDE the arg_scope that carries the regulariser (reference :62): no L2 term on its weights
    net = G.fully_connected(net, cell_width * cell_height * (classes + boxes_per_cell * 5), scope='%s/fc' % scope, activation=False)
    return scope, net


TINY_DOWNSAMPLING = (2 ** 6, 2 ** 6)

"""Graph functions of the YOLOv2 family (reference model/yolo2/function.py)."""


def reorg(net, stride=2, name='reorg'):
    """reference model/yolo2/function.py:22-29: stride-2 space-to-depth in tf.space_to_depth order (not Darknet's),
    out[b, y, x, (sy*2+sx)*C + c] = in[b, 2y+sy, 2x+sx, c].  Adds the graph node; csrc/elementwise.hip reorg_kernel executes it and
    writes straight into the concat buffer the result is re-homed in."""
    assert stride == 2 and net.h % 2 == 0 and net.w % 2 == 0
    g = net.graph
    out = g.tensor(name, net.h // 2, net.w // 2, net.c * 4)
    g.add({'kind': 'reorg', 'name': name, 'inputs': [net], 'x': net, 'out': out})
    return out

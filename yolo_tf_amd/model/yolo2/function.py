"""reorg (reference model/yolo2/function.py:22-29): stride-2 space-to-depth,
out[b, y, x, (sy*2+sx)*C + c] = in[b, 2y+sy, 2x+sx, c]  (tf.space_to_depth order, not Darknet's).
Executed by csrc/elementwise.hip reorg_kernel, writing straight into the concat buffer."""
from ...graph import reorg  # noqa: F401

"""YOLOv2 model family: ``Model`` (decode head), ``Objectives`` (anchor loss), ``Builder`` (plugin
loader + loss weighting) -- the surface of reference model/yolo2/__init__.py:28-119.

The reference builds these as TF sub-graphs; here they are thin descriptors over the traced
``graph.Graph``.  The arithmetic runs in csrc/head.hip: ``yolo2_head_decode`` for the detection
block (:50-56) and ``yolo2_loss`` for decode + Objectives + their gradient in one launch."""
import configparser  # noqa: F401  (kept: callers catch configparser errors as with the reference)
import os

import numpy as np

from ... import graph as G
from ... import utils
from .. import yolo
from . import inference

OBJECTIVE_KEYS = ('iou_best', 'iou_normal', 'coords', 'prob')   # insertion order of reference :90-94


class Model(object):
    """Per-cell / per-anchor decode of the network output [B, cell_h, cell_w, A*(5+C)], channel order
    per anchor ``[iou, x, y, w, h, cls...]`` (reference :32-49)."""

    def __init__(self, net, classes, anchors, training=False):
        self.cell_height, self.cell_width = net.h, net.w
        self.inputs = net
        self.classes = classes
        self.anchors = np.asarray(anchors, np.float32).reshape(-1, 2)
        self.training = training
        assert net.c == len(self.anchors) * (5 + classes)
        self._session = None

    @property
    def cells(self):
        return self.cell_height * self.cell_width

    # The attributes the reference's callers read off the TF graph (`conf, xy_min, xy_max` detect.py:69-72; `prob, iou,
    # xy_min, wh` demo_detect.py:62) are device tensors of the bound DetectSession's LAST run, shaped like the reference's
    # [B, cells, A, ...]: `conf / xy_min / xy_max` are the decode kernel's outputs themselves, the rest are produced on
    # first access by yolo2_head_decode_attrs from the same logits.
    def bind(self, session):
        self._session = session
        return self

    def _bound(self):
        if self._session is None:
            raise AttributeError('Model attributes hold values only after DetectSession.run(); build a session first')
        return self._session

    def _view(self, t, last):
        s = self._bound()
        return t.view(s.B, self.cells, len(self.anchors), last) if last else t.view(s.B, self.cells, len(self.anchors))

    conf = property(lambda self: self._view(self._bound().conf, self.classes))
    xy_min = property(lambda self: self._view(self._bound().xy_min, 2))
    xy_max = property(lambda self: self._view(self._bound().xy_max, 2))
    iou = property(lambda self: self._view(self._bound().attrs()['iou'], 0))
    prob = property(lambda self: self._view(self._bound().attrs()['prob'], self.classes))
    xy = property(lambda self: self._view(self._bound().attrs()['xy'], 2))
    wh = property(lambda self: self._view(self._bound().attrs()['wh'], 2))


class Objectives(dict):
    """The four masked-L2 terms (reference :62-94).  Values are filled (as Python floats) by the
    training session after every step; ``labels`` is the 6-tuple
    (mask, prob, coords, offset_xy_min, offset_xy_max, areas) of utils.data.transform_labels."""

    def __init__(self, model, mask=None, prob=None, coords=None, offset_xy_min=None, offset_xy_max=None, areas=None):
        super(Objectives, self).__init__()
        self.model = model
        self.labels = (mask, prob, coords, offset_xy_min, offset_xy_max, areas)
        for key in OBJECTIVE_KEYS:
            self[key] = float('nan')


class Builder(yolo.Builder):
    family = 'yolo2'

    def __init__(self, args, config):
        section = __name__.split('.')[-1]                    # 'yolo2'
        self.args = args                                      # stored, never read (as in the reference)
        self.config = config
        with open(os.path.join(utils.get_cachedir(config), 'names'), 'r') as f:
            self.names = [line.strip() for line in f]
        self.width = config.getint(section, 'width')
        self.height = config.getint(section, 'height')
        self.anchors = utils.read_anchors(config.get(section, 'anchors'))
        self.func = getattr(inference, config.get(section, 'inference'))
        self.graph = None

    def __call__(self, data=None, training=False):
        """Traces the network.  ``data`` is an image placeholder (graph.Tensor) or None to create one
        of the configured size."""
        if data is None:
            self.graph = G.Graph()
            data = G.placeholder(self.graph, 'image', self.height, self.width)
        else:
            self.graph = data.graph
        self.training = training
        _, self.output = self.func(data, len(self.names), len(self.anchors), training=training)
        self.model = Model(self.output, len(self.names), self.anchors, training=training)
        return self.model

    def trace(self, width, height, training=False):
        """Traces the same network at another input size (multi-scale training, BASELINE configs[3]; the reference lists it as
        future work, README.md:87): returns (graph, Model) without touching the builder's own graph / model."""
        dw, dh = getattr(inference, self.config.get(__name__.split('.')[-1], 'inference').upper() + '_DOWNSAMPLING')
        assert width % dw == 0 and height % dh == 0, 'input size must be a multiple of the downsampling (%d, %d)' % (dw, dh)
        graph = G.Graph()
        data = G.placeholder(graph, 'image', height, width)
        _, output = self.func(data, len(self.names), len(self.anchors), training=training)
        return graph, Model(output, len(self.names), self.anchors, training=training)

    def create_objectives(self, labels=(None,) * 6):
        section = __name__.split('.')[-1]
        self.objectives = Objectives(self.model, *labels)
        self.hparam = {key: self.config.getfloat(section + '_hparam', key) for key in self.objectives}
        return self.objectives

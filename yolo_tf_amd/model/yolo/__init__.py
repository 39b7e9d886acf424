"""YOLO (v1) model family: ``calc_cell_xy``, ``Model``, ``Objectives``, ``Builder`` -- the surface of reference
model/yolo/__init__.py:29-123.  As in ``model.yolo2`` these are descriptors over the traced ``graph.Graph``; the arithmetic is
csrc/yolo1.hip (``yolo1_head_decode`` for the detection block :56-62, ``yolo1_loss`` for Model + Objectives + their gradient)."""
import os

import numpy as np

from ... import graph as G
from ... import utils

OBJECTIVE_KEYS = ('iou_best', 'iou_normal', 'coords', 'prob')


def calc_cell_xy(cell_height, cell_width, dtype=np.float32):
    """cell_base[y, x, :] = [x, y]; flat cell index = y * cell_width + x (reference :29-34)."""
    xs, ys = np.meshgrid(np.arange(cell_width), np.arange(cell_height))
    return np.stack([xs, ys], axis=-1).astype(dtype)


class Builder(object):
    """Interface of a model family: __call__(data, training) traces the network, create_objectives(labels) attaches the loss.
    ``model.yolo2.Builder`` derives from it; instantiated directly it is the YOLO (v1) builder (reference :103-123)."""
    family = 'yolo'

    def __init__(self, args, config):
        section = 'yolo'
        self.args = args
        self.config = config
        with open(os.path.join(utils.get_cachedir(config), 'names'), 'r') as f:
            self.names = [line.strip() for line in f]
        self.width = config.getint(section, 'width')
        self.height = config.getint(section, 'height')
        self.boxes_per_cell = config.getint(section, 'boxes_per_cell')
        from . import inference
        self.func = getattr(inference, config.get(section, 'inference'))
        self.graph = None

    def __call__(self, data=None, training=False):
        if data is None:
            self.graph = G.Graph()
            data = G.placeholder(self.graph, 'image', self.height, self.width)
        else:
            self.graph = data.graph
        self.training = training
        _scope, self.output = self.func(data, len(self.names), self.boxes_per_cell, training=training)
        self.model = Model(self.output, _scope, len(self.names), self.boxes_per_cell, training=training)
        return self.model

    def create_objectives(self, labels=(None,) * 6):
        self.objectives = Objectives(self.model, *labels)
        self.hparam = {key: self.config.getfloat('yolo_hparam', key) for key in self.objectives}
        return self.objectives


class Model(object):
    """Split of the flat network output [B, cells*C + cells*boxes*5] (reference :37-66): ``prob`` [B, cells, 1, C], and per box
    ``iou``, ``offset_xy``, ``wh01_sqrt_base`` -- all linear.  The attribute tensors (``conf, xy_min, xy_max``) are the bound
    DetectSession's buffers, as for ``model.yolo2.Model``."""

    def __init__(self, net, scope, classes, boxes_per_cell, training=False):
        self.cell_height, self.cell_width = net.graph.cells_hw
        self.inputs = net
        self.scope = scope
        self.classes = classes
        self.boxes_per_cell = boxes_per_cell
        self.training = training
        assert net.c == self.cells * (classes + boxes_per_cell * 5)
        self._session = None

    @property
    def cells(self):
        return self.cell_height * self.cell_width

    def bind(self, session):
        self._session = session
        return self

    def _view(self, t, last):
        if self._session is None:
            raise AttributeError('Model attributes hold values only after DetectSession.run()')
        return t.view(self._session.B, self.cells, self.boxes_per_cell, last)

    conf = property(lambda self: self._view(self._session.conf if self._session else None, self.classes))
    xy_min = property(lambda self: self._view(self._session.xy_min if self._session else None, 2))
    xy_max = property(lambda self: self._view(self._session.xy_max if self._session else None, 2))


class Objectives(dict):
    """The four masked-L2 terms (reference :69-100); values are filled by the training session after every step."""

    def __init__(self, model, mask=None, prob=None, coords=None, offset_xy_min=None, offset_xy_max=None, areas=None):
        super(Objectives, self).__init__()
        self.model = model
        self.labels = (mask, prob, coords, offset_xy_min, offset_xy_max, areas)
        for key in OBJECTIVE_KEYS:
            self[key] = float('nan')

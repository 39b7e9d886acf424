"""The pieces of the reference's ``model.yolo`` package that the YOLOv2 path uses:
``calc_cell_xy`` (model/yolo/__init__.py:29-34) and the ``Builder`` base class (:103-123).
The YOLOv1 network itself (FC head, no anchors) is outside the hot path."""
import numpy as np


def calc_cell_xy(cell_height, cell_width, dtype=np.float32):
    """cell_base[y, x, :] = [x, y]; flat cell index = y * cell_width + x."""
    xs, ys = np.meshgrid(np.arange(cell_width), np.arange(cell_height))
    return np.stack([xs, ys], axis=-1).astype(dtype)


class Builder(object):
    """Interface of a model family: __call__(data, training) traces the network,
    create_objectives(labels) attaches the loss."""

    def __call__(self, data, training=False):
        raise NotImplementedError

    def create_objectives(self, labels):
        raise NotImplementedError

"""Non-max suppression with the reference's call signature and in-place contract
(utils/postprocess.py:39-51), executed by the gfx950 kernel csrc/nms.hip -- there is no host
implementation here."""
import numpy as np
import torch

from .. import ops


def non_max_suppress_device(conf, xy_min, xy_max, threshold, threshold_iou, order=None, ws=None):
    """Batched, on device.  conf [B,N,C] f32 (updated in place), xy_min/xy_max [B,N,2] f32.
    Returns ``order`` [B,N] int32: box indices in the order of the reference's returned list."""
    B, N, C = conf.shape
    assert conf.is_cuda and conf.dtype == torch.float32 and conf.is_contiguous()
    if order is None:
        order = torch.empty(B, N, dtype=torch.int32, device=conf.device)
    if ws is None:
        ws = torch.empty(B * N * C, dtype=torch.int32, device=conf.device)
    ops.nms(conf, xy_min.contiguous(), xy_max.contiguous(), order, ws, B, N, C, float(threshold), float(threshold_iou))
    return order


def non_max_suppress(conf, xy_min, xy_max, threshold, threshold_iou):
    """Drop-in for the reference: NumPy ``conf [cells,A,C]``, ``xy_min/xy_max [cells,A,2]``;
    ``conf`` is mutated in place; returns the list of ``(conf_row, xy_min, xy_max)`` triples
    (views into the caller's arrays) in the reference's order.  The consumer keeps a box when
    ``conf_row[argmax] > threshold`` (detect.py:78-80)."""
    cells, a, classes = conf.shape
    n = cells * a
    assert conf.dtype == np.float32 and conf.flags['C_CONTIGUOUS'], 'conf must be a contiguous float32 array (it is updated in place)'
    assert not np.isnan(xy_min).any() and not np.isnan(xy_max).any()     # the reference's iou() asserts
    assert np.all(xy_min <= xy_max)
    dconf = torch.from_numpy(conf.reshape(1, n, classes)).cuda()
    dmin = torch.from_numpy(np.ascontiguousarray(xy_min, np.float32).reshape(1, n, 2)).cuda()
    dmax = torch.from_numpy(np.ascontiguousarray(xy_max, np.float32).reshape(1, n, 2)).cuda()
    order = non_max_suppress_device(dconf, dmin, dmax, threshold, threshold_iou)
    conf.reshape(n, classes)[...] = dconf[0].cpu().numpy()
    order = order[0].cpu().numpy()
    cf, mn, mx = conf.reshape(n, classes), xy_min.reshape(n, 2), xy_max.reshape(n, 2)
    return [(cf[i], mn[i], mx[i]) for i in order]

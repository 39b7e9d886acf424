"""Detect-side image preprocessing (reference utils/preprocess.py:23-25, detect.py:33-38), run by
csrc/elementwise.hip image kernels."""
import numpy as np
import torch

from .. import ops


def _prep(image, mode):
    image = np.ascontiguousarray(image, np.float32)
    h, w, c = image.shape
    assert c == 3
    out = torch.zeros(h * w * 8, dtype=torch.float32, device='cuda')
    ws = torch.zeros(ops.workspace_bytes('image_prep', 1) // 8, dtype=torch.float64, device='cuda')
    ops.image_prep(torch.from_numpy(image).cuda(), out, ws, 1, h * w, mode)
    return out.reshape(h, w, 8)[..., :3].cpu().numpy()


def per_image_standardization(image):
    """(x - mean) / max(std, 1/sqrt(N)) over the whole image."""
    return _prep(image, 0)


def darknet(image):
    return _prep(image, 1)

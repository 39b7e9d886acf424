"""Label layout for the YOLOv2 loss (host side).  ``transform_labels`` produces the six per-cell
tensors the reference's input pipeline feeds ``Objectives`` (utils/data/__init__.py:112-145): one
ground-truth box per grid cell, later objects overwrite earlier ones in the same cell except for
``prob`` which becomes multi-hot.  ``synthetic_batch`` is the seeded generator SURVEY 8(d) specifies
for benchmarks (K~U{1..6} boxes per image)."""
import numpy as np

LABEL_KEYS = ('mask', 'prob', 'coords', 'offset_xy_min', 'offset_xy_max', 'areas')


def transform_labels(objects_class, objects_coord, classes, cell_width, cell_height, dtype=np.float32):
    """objects_coord [K,4] = (xmin, ymin, xmax, ymax) normalised to [0,1]."""
    cells = cell_width * cell_height
    mask = np.zeros((cells, 1), dtype)
    prob = np.zeros((cells, 1, classes), dtype)
    coords = np.zeros((cells, 1, 4), dtype)
    xy_min = np.zeros((cells, 1, 2), dtype)
    xy_max = np.zeros((cells, 1, 2), dtype)
    objects_class = np.asarray(objects_class)
    box = np.asarray(objects_coord).reshape(-1, 4)
    assert len(objects_class) == len(box)
    if len(box):
        grid = np.array([cell_width, cell_height], box.dtype)    # stay in the coordinate dtype (f32), like the reference
        centre = grid * (box[:, 0:2] + box[:, 2:4]) / 2           # box centre in cell units
        cell = np.floor(centre)
        offset = centre - cell
        size = box[:, 2:4] - box[:, 0:2]                          # normalised w, h
        index = (cell[:, 1] * cell_width + cell[:, 0]).astype(int)
        half = size / 2 * grid
        mask[index, 0] = 1
        prob[index, 0, objects_class] = 1
        coords[index, 0, 0:2] = offset
        coords[index, 0, 2:4] = np.sqrt(size)
        xy_min[index, 0] = offset - half
        xy_max[index, 0] = offset + half
    extent = xy_max - xy_min
    assert np.all(extent >= 0)
    return mask, prob, coords, xy_min, xy_max, extent[..., 0] * extent[..., 1]


def synthetic_batch(batch, classes, cell_width, cell_height, seed):
    """Seeded synthetic labels: per image K~U{1..6}, class~U, centre~U(.05,.95)^2, w,h~U(.05,.6) clipped."""
    rng = np.random.RandomState(seed)
    per_image = []
    for _ in range(batch):
        k = rng.randint(1, 7)
        centre = rng.uniform(0.05, 0.95, (k, 2))
        size = rng.uniform(0.05, 0.6, (k, 2))
        box = np.clip(np.concatenate([centre - size / 2, centre + size / 2], 1), 0, 1).astype(np.float32)
        per_image.append(transform_labels(rng.randint(0, classes, k), box, classes, cell_width, cell_height))
    return tuple(np.stack([p[i] for p in per_image]) for i in range(6))

"""Input pipeline measurement: images/s of yolo2_augment_images + yolo2_transform_labels for a batch of 16
decoded 500x375 images -> 416x416 with every augmentation branch taken, against the algorithmic HBM bytes, and the
NumPy oracle (oracle/yolo2_ref.augment_image) timed on the host beside it."""
import os, sys, time, configparser
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from yolo_tf_amd import ops
from yolo_tf_amd.utils import augment as A
from oracle import yolo2_ref as R

B, W, H, classes, cw, ch = 16, 416, 416, 20, 13, 13
rng = np.random.RandomState(0)
images = [rng.randint(0, 256, (375, 500, 3)).astype(np.uint8) for _ in range(64)]
objects = []
for _ in images:
    k = rng.randint(1, 7)
    x0, y0 = rng.uniform(0, 250, k), rng.uniform(0, 180, k)
    objects.append((rng.randint(0, classes, k), np.stack([x0, y0, x0 + rng.uniform(20, 240, k), y0 + rng.uniform(20, 180, k)], 1)))
ini = configparser.ConfigParser()
ini.read(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'config.ini'))
cfg = A.AugmentConfig(ini)
cfg.full_probability = cfg.probability = 1.0          # every branch taken: worst case
cfg.grayscale_probability = 0.0
pipe = A.DeviceInputPipeline(images, objects, B, W, H, classes, cw, ch, config=cfg, seed=1)
labels = [torch.zeros(*s, device='cuda') for s in ((B, cw * ch, 1), (B, cw * ch, 1, classes), (B, cw * ch, 1, 4), (B, cw * ch, 1, 2), (B, cw * ch, 1, 2), (B, cw * ch, 1))]
batches = [pipe.assemble(pipe.sample()) for _ in range(8)]
for b in batches[:3]:
    pipe.launch(*b, labels)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    for b in batches:
        pipe.launch(*b, labels)
e.record()
torch.cuda.synchronize()
ms = a.elapsed_time(e) / 40
src_bytes = sum(int(p.crop_w) * int(p.crop_h) * 3 for p in batches[0][0])
out_bytes = B * H * W * 3 * 4
t0 = time.time()
for b in batches:
    pipe.assemble(pipe.sample())
host_ms = (time.time() - t0) / len(batches) * 1e3
print('device: %.3f ms per batch of %d (incl. parameter upload) = %.0f img/s; algorithmic bytes %.1f MB (source crop read once x2 passes + f32 out) -> %.2f TB/s; host draw + box transforms %.3f ms per batch'
      % (ms, B, B / ms * 1e3, (2 * src_bytes + out_bytes) / 1e6, (2 * src_bytes + out_bytes) / ms / 1e9, host_ms))
# kernels alone (parameters already on the device)
arr, cls, box, first = batches[0]
params = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
for flagset, name in ((None, 'all branches'), (0, 'resize only'), (A.CONTRAST, 'contrast only'), (A.NOISE, 'noise only'), (A.SATURATION | A.HUE, 'saturation+hue')):
    arr2 = type(arr)()
    for d, s_ in zip(arr2, arr):
        for f_, _t in d._fields_:
            setattr(d, f_, getattr(s_, f_))
        if flagset is not None:
            d.flags = flagset
    pd = torch.frombuffer(bytearray(bytes(arr2)), dtype=torch.uint8).cuda()
    anyc = any(x.flags & A.CONTRAST for x in arr2)
    for _ in range(3):
        ops.augment_images(pipe.src, pd, pipe.ws, pipe.out, B, H, W, anyc)
    torch.cuda.synchronize()
    a.record()
    for _ in range(20):
        ops.augment_images(pipe.src, pd, pipe.ws, pipe.out, B, H, W, anyc)
    e.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(e) / 20
    print('  augment kernels, %-16s %.1f us per batch  (%.2f TB/s of f32 output alone)' % (name + ':', t * 1e3, out_bytes / t / 1e9))
# CPU: the oracle on one image with the same branches
p = dict(crop=(10, 8, 400, 300), flip=True, brightness=20.0, saturation=1.2, hue=0.01, contrast=1.1, noise=None, gray=False)
t0 = time.time(); n = 0
while time.time() - t0 < 5:
    R.augment_image(images[n % 64], p, W, H); n += 1
print('cpu oracle (NumPy, 1 core): %.1f img/s' % (n / (time.time() - t0)))

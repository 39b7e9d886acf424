"""Would a hipGraph of the WHOLE training step be faster than enqueueing it?  Captures TrainSession.step (both streams, events and all) into one graph and
replays it.  The captured step freezes the host-computed Adam step size -- a timing probe, not a training loop.  usage: python scripts/graph_step_probe.py"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from yolo_tf_amd.session import TrainSession
from yolo_tf_amd.utils import data
B, size, names = int(os.environ.get('B', 16)), 416, 20
builder, cfg = bench.make_builder('darknet', names, size, True, tempfile.mkdtemp())
sess = TrainSession(builder, B, dtype='bf16', optimizer='adam', learning_rate=1e-6, seed=0)
sess.async_errors = None
gen = torch.Generator(device='cuda').manual_seed(1234)
images = torch.rand(B, size, size, 3, device='cuda', generator=gen) * 255.0
sess.upload_labels(data.synthetic_batch(B, names, size // 32, size // 32, seed=4321))


def eager(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        sess.step(images)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(10):
    sess.step(images)
print('eager: %.3f ms/step' % eager(60), flush=True)
g = torch.cuda.CUDAGraph()
cap = torch.cuda.Stream()
cap.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(cap):
    with torch.cuda.graph(g, stream=cap):
        sess.step(images)
        if os.environ.get('JOIN') and getattr(sess.engine, 'side_stream', None) is not None:
            cap.wait_stream(sess.engine.side_stream)
torch.cuda.synchronize()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(60):
        g.replay()
    torch.cuda.synchronize()
    print('graph replay: %.3f ms/step' % ((time.perf_counter() - t0) / 60 * 1e3), flush=True)
    print('eager: %.3f ms/step' % eager(60), flush=True)

"""Where the HOST time of a training step goes (cProfile over K enqueued steps; the GPU runs behind).  python scripts/host_profile.py [batch]"""
import cProfile, os, pstats, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from yolo_tf_amd.session import TrainSession
from yolo_tf_amd.utils import data
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
b, cfg = bench.make_builder('darknet', 20, 416, True, tempfile.mkdtemp())
sess = TrainSession(b, B, dtype='bf16', optimizer='adam', learning_rate=1e-4, config=cfg, seed=0)
images = torch.rand(B, 416, 416, 3, device='cuda') * 255.0
sess.upload_labels(data.synthetic_batch(B, 20, 13, 13, seed=1))
for _ in range(5):
    sess.step(images)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    sess.step(images)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)

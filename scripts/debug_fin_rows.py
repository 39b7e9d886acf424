"""Which BN layers take the folded-finalisation consumers at a given batch (and why not): prints (rows, C, supported) per producer launch.
    python scripts/debug_fin_rows.py [batch] [names]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from yolo_tf_amd import ops
from yolo_tf_amd.session import TrainSession
from yolo_tf_amd.utils import data
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
names = int(sys.argv[2]) if len(sys.argv) > 2 else 80
b, cfg = bench.make_builder('darknet', names, 416, True, tempfile.mkdtemp())
sess = TrainSession(b, B, dtype='bf16', optimizer='adam', learning_rate=1e-4, config=cfg, seed=0)
images = torch.rand(B, 416, 416, 3, device='cuda') * 255.0
sess.upload_labels(data.synthetic_batch(B, names, 13, 13, seed=1))
sess.step(images)
orig = ops.bn_fin_supported
def spy(rows, C, dtype):
    r = orig(rows, C, dtype)
    print('rows %4d  C %4d  supported %s   plan %s' % (rows, C, r, ops.last_conv_plan()))
    return r
ops.bn_fin_supported = spy
import yolo_tf_amd.engine as E
E.ops.bn_fin_supported = spy
sess.step(images)
torch.cuda.synchronize()

"""Layer-by-layer backward comparison engine vs oracle (f32) to locate a divergence."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import yolo2_ref as R
from bench import make_builder
from yolo_tf_amd.session import TrainSession
from yolo_tf_amd.utils import data
from yolo_tf_amd import ops

size, B, classes = int(os.environ.get('SIZE', 96)), 2, 20
b, _ = make_builder('darknet', classes, size, True, tempfile.mkdtemp())
DT = os.environ.get('DTYPE', 'f32')
Q = None if DT == 'f32' else R.bf16_round
sess = TrainSession(b, B, dtype=DT, optimizer='adam', learning_rate=1e-3, seed=3)
e = sess.engine
scope = 'yolo2_darknet/'
params = {k[len(scope):]: v for k, v in e.get_variables().items()}
rng = np.random.RandomState(0)
for k in list(params):
    if k.endswith('gamma'): params[k] = (rng.rand(*params[k].shape) + 0.5).astype(np.float32)
    if k.endswith(('beta', 'biases')): params[k] = (rng.randn(*params[k].shape) * 0.1).astype(np.float32)
e.set_variables({scope + k: v for k, v in params.items()})
cells = size // 32
images = rng.uniform(0, 255, (B, size, size, 3)).astype(np.float32)
labels = data.synthetic_batch(B, classes, cells, cells, seed=7)
sess.upload_labels(labels)
sess.forward_backward(torch.from_numpy(images).cuda())
torch.cuda.synchronize()

spec = R.darknet_spec(classes, 5)
x = np.stack([R.per_image_standardization(i) for i in images]).astype(np.float32)
net, caches = R.network_forward(spec, params, x, True, quant=Q)
m = R.model_decode(net, classes, b.anchors, True)
hp = {'prob': 1., 'iou_best': 5., 'iou_normal': 1., 'coords': 1.}
obj, aux = R.objectives(m, labels)
dnet = R.loss_backward(m, labels, aux, hp, classes)
if Q: dnet = Q(dnet)

def rel(a, r): return float(np.abs(a - r).max() / (np.abs(r).max() + 1e-30))
def gpu_act(t, grad=False):
    buf, ld = (e.gact if grad else e.act)[t]
    n = B * t.h * t.w
    return buf[:n * ld].float().cpu().numpy().reshape(n, ld)[:, :t.c].reshape(B, t.h, t.w, t.c) if ld * n <= buf.numel() else None

gops = [o for o in e.graph.ops]
conv_ops = {o['name'].split('/')[-1]: o for o in gops if o['kind'] == 'conv'}
grads_gpu = {k[len(scope):]: v for k, v in e.get_gradients().items()}
dmark = None
print('%-8s %10s %10s %10s %10s %10s | fwd: %10s %10s %10s' % ('layer', 'd_out', 'dW', 'dgamma', 'dbeta', 'd_in', 'y', 'mean', 'var'))
for op, cache in zip(reversed(spec), reversed(caches['ops'])):
    if op[0] == 'conv':
        _, name, k, cout, bn = op
        _, _, xin, y, mean, var, z = cache
        gop = conv_ops[name]
        # slice-aware read of d(out)
        t = gop['out']
        buf, ld = e.gact[t]
        n = B * t.h * t.w
        flat = buf.float().cpu().numpy()
        d_out_g = np.stack([flat[i * ld:i * ld + t.c] for i in range(n)]).reshape(B, t.h, t.w, t.c)
        r_dout = rel(d_out_g, dnet)
        w = params[name + '/weights'] if not Q else Q(params[name + '/weights'])
        if bn:
            dz = R.leaky_relu_grad(z, dnet)
            dy, dg, db = R.bn_train_bwd(y, mean, var, params[name + '/BatchNorm/gamma'], dz)
            if Q: dy = Q(dy)
            r_dg, r_db = rel(grads_gpu[name + '/BatchNorm/gamma'], dg), rel(grads_gpu[name + '/BatchNorm/beta'], db)
            st = e.conv[gop['name']]
            ry = rel(gpu_act(gop['y']), y); rm = rel(st['mean'].cpu().numpy(), mean); rv = rel(st['var'].cpu().numpy(), var)
        else:
            dy = dnet; r_dg = r_db = ry = rm = rv = float('nan')
        dW = R.conv2d_wgrad(xin, dy, k, k)
        r_dw = rel(grads_gpu[name + '/weights'], dW)
        dnet = R.conv2d_dgrad(dy, w)
        if Q: dnet = Q(dnet)
        r_din = float('nan')
        if gop['x'] in e.gact and name != 'conv13':
            tx = gop['x']; bufx, ldx = e.gact[tx]; nx = B * tx.h * tx.w; fx = bufx.float().cpu().numpy()
            d_in_g = np.stack([fx[i * ldx:i * ldx + tx.c] for i in range(nx)]).reshape(B, tx.h, tx.w, tx.c)
            r_din = rel(d_in_g, dnet) if name not in ('conv13',) else float('nan')
        print('%-8s %10.2e %10.2e %10.2e %10.2e %10.2e | %15.2e %10.2e %10.2e  minvar %.2e' % (name, r_dout, r_dw, r_dg, r_db, r_din, ry, rm, rv, float(var.min()) if bn else 0))
    elif op[0] == 'pool':
        dnet = R.max_pool_grad(cache[2], dnet, cache[1])
    elif op[0] == 'mark':
        dnet = dnet + dmark if not Q else Q(dnet + dmark)
    elif op[0] == 'reorg_concat':
        cr = cache[1]
        dmark = R.reorg_grad(dnet[..., :cr]); dnet = dnet[..., cr:]

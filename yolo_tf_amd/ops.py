"""Thin tensor-level wrappers over the C-ABI kernels (torch tensors are only device memory +
stream plumbing here).  Used by the engine and by the per-kernel parity tests."""
import ctypes

import torch

from . import _lib
from ._lib import call, ptr, dtype_code


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_raw_device = getattr(torch._C, '_cuda_getDevice', None)


def _stream():
    """The current HIP stream of the current device as a raw handle.  torch.cuda.current_stream() builds a Stream object through four
    Python layers (device-index parsing, is_available(), lazy-init checks): ~8 us per call, as much as the kernel launch itself and 28 %
    of the host time of a training step (scripts/host_profile.py); the two C accessors cost ~0.3 us together."""
    if _raw_stream is None or _raw_device is None:
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    return ctypes.c_void_p(_raw_stream(_raw_device()))


def pad8(c):
    return (c + 7) // 8 * 8


def conv2d(P, F, bias, O, B, H, W, Cp, ldp, Nf, ldo, ksize):
    call('yolo2_conv2d', ptr(P), ptr(F), ptr(bias), ptr(O), B, H, W, Cp, ldp, Nf, ldo, ksize, dtype_code(P.dtype), _stream())


def conv2d_ws(P, F, bias, O, ws, B, H, W, Cp, ldp, Nf, ldo, ksize):
    call('yolo2_conv2d_ws', ptr(P), ptr(F), ptr(bias), ptr(O), ptr(ws), ws.numel() * ws.element_size(), B, H, W, Cp, ldp, Nf, ldo, ksize,
         dtype_code(P.dtype), _stream())


def conv2d_bias_leaky(P, F, bias, O, ws, B, H, W, Cp, ldp, Nf, ldo, ksize, alpha):
    call('yolo2_conv2d_bias_leaky', ptr(P), ptr(F), ptr(bias), ptr(O), ptr(ws), ws.numel() * ws.element_size(), B, H, W, Cp, ldp, Nf, ldo, ksize,
         alpha, dtype_code(P.dtype), _stream())


def conv2d_bn(P, F, O, ws, B, H, W, Cp, ldp, Nf, ldo, ksize, shift, bn_part):
    call('yolo2_conv2d_bn', ptr(P), ptr(F), ptr(O), ptr(ws), ws.numel() * ws.element_size(), B, H, W, Cp, ldp, Nf, ldo, ksize,
         ptr(shift), ptr(bn_part), dtype_code(P.dtype), _stream())


def conv2d_dgrad_bn(dY, F, dX, ws, B, H, W, Cp, ldp, Nf, ldo, ksize, Yprev, mean, var, gamma, beta, dgamma, dbeta, bn_part, red_ws, eps, alpha):
    """-> True when the sums are still in ``bn_part`` (finish with ``bn_part_to_grads``)."""
    pending = ctypes.c_int(0)
    call('yolo2_conv2d_dgrad_bn', ptr(dY), ptr(F), ptr(dX), ptr(ws), ws.numel() * ws.element_size(), B, H, W, Cp, ldp, Nf, ldo, ksize,
         ptr(Yprev), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(dgamma), ptr(dbeta), ptr(bn_part), ptr(red_ws), eps, alpha,
         ctypes.byref(pending), dtype_code(dY.dtype), _stream())
    return bool(pending.value)


def conv2d_dgrad_bn_fuses(B, H, W, Nf, ksize, dtype):
    """Would ``conv2d_dgrad_bn`` take the producer's sums from its epilogue for this data gradient?  (False: the launch rule prefers a plain data
    gradient there; the caller schedules ``bn_leaky_bwd_reduce_part`` + ``bn_leaky_bwd_apply_fin`` itself.)"""
    lib = _lib.load()
    if 'yolo2_conv2d_dgrad_bn_fuses' in _lib.MISSING:       # (an earlier round's library, YOLO2_LIB_BASELINE=1: it always fused)
        return True
    return bool(lib.yolo2_conv2d_dgrad_bn_fuses(B, H, W, Nf, ksize, dtype_code(dtype)))


def filter_prep_blocks(ksize, ldcin, ldcout):
    """Workgroups one layer takes in ``filter_prep_batch`` / ``adam_filter_prep`` (the layer's ``first_block`` arithmetic)."""
    lib = _lib.load()
    if 'yolo2_filter_prep_blocks' in _lib.MISSING:       # (an earlier round's library, YOLO2_LIB_BASELINE=1: 64 x 64 tiles)
        return ksize * ksize * ((ldcin + 63) // 64) * ((ldcout + 63) // 64)
    return int(lib.yolo2_filter_prep_blocks(ksize, ldcin, ldcout))


def bn_part_to_grads(bn_part, C, dgamma, dbeta):
    call('yolo2_bn_part_to_grads', ptr(bn_part), C, ptr(dgamma), ptr(dbeta), _stream())


def bn_finalize(bn_part, shift, M, C, mean, var, mm, mv, decay):
    call('yolo2_bn_finalize', ptr(bn_part), ptr(shift), M, C, ptr(mean), ptr(var), ptr(mm), ptr(mv), decay, _stream())


def conv2d_wgrad_accumulates(B, H, W, Cin, ldx, Cout, ldy, ksize, dtype):
    """True when yolo2_conv2d_wgrad adds into dW with atomics for this shape (dW must be zeroed first); False when it overwrites."""
    return bool(_lib.load().yolo2_conv2d_wgrad_accumulates(B, H, W, Cin, ldx, Cout, ldy, ksize, dtype_code(dtype)))


def conv2d_wgrad(X, dY, dW, B, H, W, Cin, ldx, Cout, ldy, ksize):
    call('yolo2_conv2d_wgrad', ptr(X), ptr(dY), ptr(dW), B, H, W, Cin, ldx, Cout, ldy, ksize, dtype_code(X.dtype), _stream())


def filter_prep(Wt, Ffwd, Fdgr, ksize, Cin, ldcin, Cout, ldcout, dtype):
    call('yolo2_filter_prep', ptr(Wt), ptr(Ffwd), ptr(Fdgr), ksize, Cin, ldcin, Cout, ldcout, dtype_code(dtype), _stream())


def filter_prep_batch(descs_dev, n, total_blocks, dtype):
    call('yolo2_filter_prep_batch', ptr(descs_dev), n, total_blocks, dtype_code(dtype), _stream())


def adam_filter_prep(descs_dev, n, total_blocks, small_dev, n_small, params, grads, m, v, alpha, b1, b2, eps, gscale, dtype):
    call('yolo2_adam_filter_prep', ptr(descs_dev), n, total_blocks, ptr(small_dev), n_small, ptr(params), ptr(grads), ptr(m), ptr(v), alpha, b1, b2, eps, gscale,
         dtype_code(dtype), _stream())


def bn_stats(Y, mean, var, ws, M, C):
    call('yolo2_bn_stats', ptr(Y), ptr(mean), ptr(var), ptr(ws), M, C, dtype_code(Y.dtype), _stream())


def bn_stats_ema(Y, mean, var, mm, mv, decay, ws, M, C):
    call('yolo2_bn_stats_ema', ptr(Y), ptr(mean), ptr(var), ptr(mm), ptr(mv), decay, ptr(ws), M, C, dtype_code(Y.dtype), _stream())


def bn_ema(mm, mv, mean, var, C, decay):
    call('yolo2_bn_ema', ptr(mm), ptr(mv), ptr(mean), ptr(var), C, decay, _stream())


def bn_leaky(Y, mean, var, gamma, beta, A, M, C, lda, eps, alpha):
    call('yolo2_bn_leaky', ptr(Y), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(A), M, C, lda, eps, alpha, dtype_code(Y.dtype), _stream())


def bn_leaky_bwd_reduce(dA, ldda, Y, mean, var, gamma, beta, dgamma, dbeta, ws, M, C, eps, alpha):
    call('yolo2_bn_leaky_bwd_reduce', ptr(dA), ldda, ptr(Y), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(dgamma), ptr(dbeta), ptr(ws),
         M, C, eps, alpha, dtype_code(Y.dtype), _stream())


def bn_leaky_bwd_apply(dA, ldda, Y, mean, var, gamma, beta, dgamma, dbeta, dY, M, C, eps, alpha):
    call('yolo2_bn_leaky_bwd_apply', ptr(dA), ldda, ptr(Y), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(dgamma), ptr(dbeta), ptr(dY),
         M, C, eps, alpha, dtype_code(Y.dtype), _stream())


# ---- consumers that finalise the partial rows in their own prologue (include/yolo2_hip.h: yolo2_bn_leaky_fin & co.)
def last_bn_part_rows():
    """Partial rows used by this thread's most recent statistics-producing launch (conv2d_bn / conv2d_dgrad_bn)."""
    return int(_lib.load().yolo2_last_bn_part_rows())


def bn_fin_supported(rows, C, dtype):
    return bool(_lib.load().yolo2_bn_fin_supported(int(rows), int(C), dtype_code(dtype)))


def bn_fin_rows_limit(C, dtype, cap=1024):
    """Largest power-of-two partial-row count (<= cap) a *_fin consumer accepts for C channels; 0 if the shape never qualifies."""
    r = cap
    while r >= 1 and not bn_fin_supported(r, C, dtype):
        r //= 2
    return r


def _zero_args(zero, zero_floats):
    return (ptr(zero) if zero_floats else None), int(zero_floats)


def bn_leaky_fin(Y, bn_part, rows, shift, mean, var, mm, mv, decay, gamma, beta, A, M, C, lda, eps, alpha, zero=None, zero_floats=0):
    z, zn = _zero_args(zero, zero_floats)
    call('yolo2_bn_leaky_fin', ptr(Y), ptr(bn_part), rows, ptr(shift), ptr(mean), ptr(var), ptr(mm), ptr(mv), decay, ptr(gamma), ptr(beta), ptr(A),
         M, C, lda, eps, alpha, z, zn, dtype_code(Y.dtype), _stream())


def bn_leaky_pool_fin(Y, bn_part, rows, shift, mean, var, mm, mv, decay, gamma, beta, P, idx, B, H, W, C, ldp, eps, alpha, zero=None, zero_floats=0, ymax=None, a_full=None):
    z, zn = _zero_args(zero, zero_floats)
    call('yolo2_bn_leaky_pool_fin', ptr(Y), ptr(bn_part), rows, ptr(shift), ptr(mean), ptr(var), ptr(mm), ptr(mv), decay, ptr(gamma), ptr(beta),
         ptr(P), ptr(idx), ptr(ymax), ptr(a_full), B, H, W, C, ldp, eps, alpha, z, zn, dtype_code(Y.dtype), _stream())


def bn_leaky_bwd_apply_fin(dA, ldda, Y, mean, var, gamma, beta, part, rows, plane_stride, dgamma, dbeta, dY, M, C, eps, alpha, zero=None, zero_floats=0):
    z, zn = _zero_args(zero, zero_floats)
    call('yolo2_bn_leaky_bwd_apply_fin', ptr(dA), ldda, ptr(Y), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(part), rows, plane_stride,
         ptr(dgamma), ptr(dbeta), ptr(dY), M, C, eps, alpha, z, zn, dtype_code(Y.dtype), _stream())


def bn_leaky_pool_bwd_apply_fin(dP, lddp, idx, Y, mean, var, gamma, beta, part, rows, plane_stride, dgamma, dbeta, dY, B, H, W, C, eps, alpha,
                                zero=None, zero_floats=0):
    z, zn = _zero_args(zero, zero_floats)
    call('yolo2_bn_leaky_pool_bwd_apply_fin', ptr(dP), lddp, ptr(idx), ptr(Y), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(part), rows, plane_stride,
         ptr(dgamma), ptr(dbeta), ptr(dY), B, H, W, C, eps, alpha, z, zn, dtype_code(Y.dtype), _stream())


def first_layer_wgrad_bn(X, Y, dP, lddp, idx, mean, var, gamma, beta, part, rows, plane_stride, dgamma, dbeta, dW, B, H, W, Cin, eps, alpha,
                         zero=None, zero_floats=0):
    """Image layer: ``bn_leaky_pool_bwd_apply_fin`` + ``conv2d_wgrad`` in one launch (the layer's output gradient never reaches HBM)."""
    z, zn = _zero_args(zero, zero_floats)
    call('yolo2_first_layer_wgrad_bn', ptr(X), ptr(Y), ptr(dP), lddp, ptr(idx), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(part), rows, plane_stride,
         ptr(dgamma), ptr(dbeta), ptr(dW), B, H, W, Cin, eps, alpha, z, zn, dtype_code(Y.dtype), _stream())


def bn_leaky_bwd_reduce_part(dA, ldda, Y, mean, var, gamma, beta, ws, rows_limit, M, C, eps, alpha):
    """-> number of partial rows left in ``ws`` ([2][rows][C] f32)."""
    rows = ctypes.c_int(0)
    call('yolo2_bn_leaky_bwd_reduce_part', ptr(dA), ldda, ptr(Y), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(ws), ctypes.byref(rows), rows_limit, M, C,
         eps, alpha, dtype_code(Y.dtype), _stream())
    return rows.value


def bn_leaky_pool_bwd_reduce_part(dP, lddp, idx, Y, mean, var, gamma, beta, ws, rows_limit, B, H, W, C, eps, alpha):
    rows = ctypes.c_int(0)
    call('yolo2_bn_leaky_pool_bwd_reduce_part', ptr(dP), lddp, ptr(idx), ptr(Y), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(ws), ctypes.byref(rows),
         rows_limit, B, H, W, C, eps, alpha, dtype_code(Y.dtype), _stream())
    return rows.value


def bn_leaky_pool(Y, mean, var, gamma, beta, P, idx, B, H, W, C, ldp, eps, alpha):
    call('yolo2_bn_leaky_pool', ptr(Y), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(P), ptr(idx), B, H, W, C, ldp, eps, alpha,
         dtype_code(Y.dtype), _stream())


def bn_leaky_pool_bwd_reduce(dP, lddp, idx, Y, mean, var, gamma, beta, dgamma, dbeta, ws, B, H, W, C, eps, alpha):
    call('yolo2_bn_leaky_pool_bwd_reduce', ptr(dP), lddp, ptr(idx), ptr(Y), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(dgamma), ptr(dbeta),
         ptr(ws), B, H, W, C, eps, alpha, dtype_code(Y.dtype), _stream())


def bn_leaky_pool_bwd_apply(dP, lddp, idx, Y, mean, var, gamma, beta, dgamma, dbeta, dY, B, H, W, C, eps, alpha):
    call('yolo2_bn_leaky_pool_bwd_apply', ptr(dP), lddp, ptr(idx), ptr(Y), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(dgamma), ptr(dbeta),
         ptr(dY), B, H, W, C, eps, alpha, dtype_code(Y.dtype), _stream())


def maxpool_fwd(A, P, B, H, W, C, stride):
    call('yolo2_maxpool_fwd', ptr(A), ptr(P), B, H, W, C, stride, dtype_code(A.dtype), _stream())


def maxpool_bwd(A, dP, dA, B, H, W, C, stride):
    call('yolo2_maxpool_bwd', ptr(A), ptr(dP), ptr(dA), B, H, W, C, stride, dtype_code(A.dtype), _stream())


def maxpool_bwd_acc(A, dP, dA, B, H, W, C):
    call('yolo2_maxpool_bwd_acc', ptr(A), ptr(dP), ptr(dA), B, H, W, C, dtype_code(A.dtype), _stream())


def reorg(x, out, B, H, W, C, ldo):
    call('yolo2_reorg', ptr(x), ptr(out), B, H, W, C, ldo, dtype_code(x.dtype), _stream())


def reorg_bwd(dout, ldd, din, B, H, W, C):
    call('yolo2_reorg_bwd', ptr(dout), ldd, ptr(din), B, H, W, C, dtype_code(din.dtype), _stream())


def copy_channels(src, lds, dst, ldd, M, C):
    call('yolo2_copy_channels', ptr(src), lds, ptr(dst), ldd, M, C, dtype_code(src.dtype), _stream())


def add_inplace(dst, src, n):
    call('yolo2_add_inplace', ptr(dst), ptr(src), n, dtype_code(dst.dtype), _stream())


def bias_grad(dY, ld, dbias, ws, M, C):
    call('yolo2_bias_grad', ptr(dY), ld, ptr(dbias), ptr(ws), M, C, dtype_code(dY.dtype), _stream())


def image_prep(img, out, ws, B, HW, mode):
    call('yolo2_image_prep', ptr(img), ptr(out), ptr(ws), B, HW, mode, dtype_code(out.dtype), _stream())


def head_decode(logits, ld, anchors, conf, xy_min, xy_max, nan_flag, B, ch, cw, A, C):
    call('yolo2_head_decode', ptr(logits), ld, ptr(anchors), ptr(conf), ptr(xy_min), ptr(xy_max), ptr(nan_flag), B, ch, cw, A, C,
         dtype_code(logits.dtype), _stream())


def head_decode_attrs(logits, ld, anchors, iou, prob, xy, wh, B, ch, cw, A, C):
    call('yolo2_head_decode_attrs', ptr(logits), ld, ptr(anchors), ptr(iou), ptr(prob), ptr(xy), ptr(wh), B, ch, cw, A, C,
         dtype_code(logits.dtype), _stream())


def loss(logits, ld, anchors, labels, hparam, objectives, dlogits, ws, B, ch, cw, A, C):
    """labels: 6 device f32 tensors (mask, prob, coords, off_min, off_max, areas); hparam: 4 host floats
    in the order iou_best, iou_normal, coords, prob."""
    hp = (ctypes.c_float * 4)(*[float(v) for v in hparam])
    mask, prob, coords, omin, omax, areas = labels
    call('yolo2_loss', ptr(logits), ld, ptr(anchors), ptr(mask), ptr(prob), ptr(coords), ptr(omin), ptr(omax), ptr(areas), hp,
         ptr(objectives), ptr(dlogits), ptr(ws), B, ch, cw, A, C, dtype_code(logits.dtype), _stream())


def loss_partials(logits, ld, anchors, labels, hparam, dlogits, ws, B, ch, cw, A, C):
    """The loss kernel alone: dlogits + per-workgroup partial sums in ``ws``; ``loss_objectives`` reduces them on demand."""
    hp = (ctypes.c_float * 4)(*[float(v) for v in hparam])
    mask, prob, coords, omin, omax, areas = labels
    call('yolo2_loss_partials', ptr(logits), ld, ptr(anchors), ptr(mask), ptr(prob), ptr(coords), ptr(omin), ptr(omax), ptr(areas), hp,
         ptr(dlogits), ptr(ws), B, ch, cw, A, C, dtype_code(logits.dtype), _stream())


def loss_objectives(ws, objectives, B, ch, cw, A):
    call('yolo2_loss_objectives', ptr(ws), ptr(objectives), B, ch, cw, A, _stream())


def loss_ws_floats(B, cells, A):
    return workspace_bytes('loss', B, cells, A) // 4


def nms(conf, xy_min, xy_max, order, ws, B, N, C, thr, thr_iou):
    call('yolo2_nms', ptr(conf), ptr(xy_min), ptr(xy_max), ptr(order), ptr(ws), B, N, C, thr, thr_iou, _stream())


def adam(w, g, m, v, n, alpha, b1, b2, eps, gscale=1.0):
    call('yolo2_adam', ptr(w), ptr(g), ptr(m), ptr(v), n, alpha, b1, b2, eps, gscale, _stream())


def momentum(w, g, acc, n, lr, mom, gscale=1.0):
    call('yolo2_momentum', ptr(w), ptr(g), ptr(acc), n, lr, mom, gscale, _stream())


def sgd(w, g, n, lr, gscale=1.0):
    call('yolo2_sgd', ptr(w), ptr(g), n, lr, gscale, _stream())


def rmsprop(w, g, ms, mom, n, lr, decay, momentum_, eps, gscale=1.0):
    call('yolo2_rmsprop', ptr(w), ptr(g), ptr(ms), ptr(mom), n, lr, decay, momentum_, eps, gscale, _stream())


def adagrad(w, g, acc, n, lr, gscale=1.0):
    call('yolo2_adagrad', ptr(w), ptr(g), ptr(acc), n, lr, gscale, _stream())


def adadelta(w, g, acc, accu, n, lr, rho, eps, gscale=1.0):
    call('yolo2_adadelta', ptr(w), ptr(g), ptr(acc), ptr(accu), n, lr, rho, eps, gscale, _stream())


def ftrl(w, g, accum, linear, n, lr, lr_power, l1, l2, gscale=1.0):
    call('yolo2_ftrl', ptr(w), ptr(g), ptr(accum), ptr(linear), n, lr, lr_power, l1, l2, gscale, _stream())


def scale(x, n, s):
    call('yolo2_scale', ptr(x), n, s, _stream())


def zero_ranges(x, ranges):
    """x[a:b] = 0 for every (a, b) in ``ranges`` (element offsets; host list)."""
    flat = (ctypes.c_long * (2 * len(ranges)))(*[int(v) for ab in ranges for v in ab])
    call('yolo2_zero_ranges', ptr(x), flat, len(ranges), _stream())


def bn_fold(W, gamma, beta, mean, var, Wf, bias, rows, C, eps):
    call('yolo2_bn_fold', ptr(W), ptr(gamma), ptr(beta), ptr(mean), ptr(var), ptr(Wf), ptr(bias), rows, C, eps, _stream())


def workspace_bytes(kind, *args):
    """Caller-owned scratch size of an entry family: 'conv2d', 'bn', 'bias_grad', 'image_prep', 'loss', 'nms', 'clip', 'augment'
    (include/yolo2_hip.h yolo2_*_workspace_bytes)."""
    return int(_lib.query('yolo2_%s_workspace_bytes' % kind, *args))


def clip_by_norm(g, seg_off, nseg, clip, ws):
    call('yolo2_clip_by_norm', ptr(g), ptr(seg_off), nseg, clip, ptr(ws), _stream())


def selftest_tr16():
    out = torch.zeros(256, dtype=torch.int16, device='cuda')
    call('yolo2_selftest_tr16', ptr(out), _stream())
    torch.cuda.synchronize()
    return out.cpu().numpy().reshape(64, 4)


CONV_PLAN_KEYS = ('BM', 'BN', 'waves', 'chunks', 'stages', 'split', 'grid_x', 'grid_y')
WGRAD_PLAN_KEYS = ('BC', 'BN', 'waves', 'pair', 'ranges', 'remap', 'blocks', 'direct')


def last_conv_plan():
    """Variant of this thread's most recent conv2d* launch (include/yolo2_hip.h yolo2_debug_last_conv_plan)."""
    out = (ctypes.c_int * 8)()
    _lib.load().yolo2_debug_last_conv_plan(out)
    return dict(zip(CONV_PLAN_KEYS, list(out)))


def noop():
    """Empty kernel on the current stream (event-bracket calibration in bench.py)."""
    call('yolo2_debug_noop', _stream())


def set_igemm_tap(mode):
    """Test hook (process-wide): 0 = per-tap 3x3 kernels only, non-zero = the ping-pong tap-fused kernel where its launch rule admits it (default)."""
    _lib.load().yolo2_debug_set_igemm_tap(int(mode))


def set_pp(grid=-1, dmapos=-1, min_steps=-1, min_share=-1):
    """Test / A-B hook for the ping-pong kernel (include/yolo2_hip.h yolo2_debug_set_pp); negative = keep."""
    _lib.load().yolo2_debug_set_pp(int(grid), int(dmapos), int(min_steps), int(min_share))


def set_pp_cost(cv):
    """Owner cost (K steps) of the ping-pong kernel's cost-balanced stream-K partition (yolo2_debug_set_pp_cost); 0 = equal K-step shares."""
    lib = _lib.load()
    if 'yolo2_debug_set_pp_cost' not in _lib.MISSING:
        lib.yolo2_debug_set_pp_cost(int(cv))


def check_async_errors():
    """Synchronises the current stream; raises RuntimeError when a stream-K tile owner gave up waiting for a partner (include/yolo2_hip.h
    yolo2_check_async_errors): errors surface, the device never hangs."""
    call('yolo2_check_async_errors', _stream())


class AsyncErrorPoll(object):
    """yolo2_async_error_snapshot without a synchronisation: ``snapshot()`` enqueues the 32-byte copy of the device's give-up counters into
    pinned host memory behind the work already on the current stream; ``pending()`` says whether a COMPLETED snapshot shows a failure (then
    ``check_async_errors()`` raises with the library's message and resets the pool).  One step of latency instead of one summary interval."""

    def __init__(self):
        self.words = torch.zeros(8, dtype=torch.int32).pin_memory()
        self.event = None

    def snapshot(self):
        _lib.load()
        if 'yolo2_async_error_snapshot' in _lib.MISSING:      # an earlier round's library in a same-box A/B (YOLO2_LIB_BASELINE=1)
            return
        call('yolo2_async_error_snapshot', self.words.data_ptr(), _stream())
        self.event = torch.cuda.Event()
        self.event.record()

    def pending(self):
        return self.event is not None and self.event.query() and bool(self.words.any().item())

    def raise_if_pending(self):
        if self.pending():
            self.event = None
            check_async_errors()
            raise RuntimeError('stream-K hand-off gave up (reported by yolo2_async_error_snapshot); convolution outputs since the last check are invalid')


def set_streamk_wait_us(us=0, unclamped=False):
    """Test hook: wait limit of the stream-K owners (0 = default 2 s) and the grid <= K-steps clamp."""
    call('yolo2_debug_set_streamk_wait_us', int(us), int(bool(unclamped)))


def last_wgrad_plan():
    out = (ctypes.c_int * 8)()
    _lib.load().yolo2_debug_last_wgrad_plan(out)
    return dict(zip(WGRAD_PLAN_KEYS, list(out)))


def set_wgrad_variant(v):
    _lib.load().yolo2_debug_set_wgrad_variant(int(v))


def augment_images(src, params_dev, ws, out, B, H, W, any_contrast):
    call('yolo2_augment_images', ptr(src), ptr(params_dev), ptr(ws), ptr(out), B, H, W, int(bool(any_contrast)), _stream())


def transform_labels(objects_class, objects_coord, first_object, mask, prob, coords, offset_xy_min, offset_xy_max, areas, B, classes,
                     cell_width, cell_height, error_flag):
    call('yolo2_transform_labels', ptr(objects_class), ptr(objects_coord), ptr(first_object), ptr(mask), ptr(prob), ptr(coords),
         ptr(offset_xy_min), ptr(offset_xy_max), ptr(areas), B, classes, cell_width, cell_height, ptr(error_flag), _stream())


# ---- YOLO (v1) family
def yolo1_loss(net, ld, labels, hparam, objectives, dnet, ws, B, ch, cw, boxes, C):
    hp = (ctypes.c_float * 4)(*[float(v) for v in hparam])
    mask, prob, coords, omin, omax, areas = labels
    call('yolo1_loss', ptr(net), ld, ptr(mask), ptr(prob), ptr(coords), ptr(omin), ptr(omax), ptr(areas), hp, ptr(objectives), ptr(dnet), ptr(ws),
         B, ch, cw, boxes, C, dtype_code(net.dtype), _stream())


def yolo1_head_decode(net, ld, conf, xy_min, xy_max, nan_flag, B, ch, cw, boxes, C):
    call('yolo1_head_decode', ptr(net), ld, ptr(conf), ptr(xy_min), ptr(xy_max), ptr(nan_flag), B, ch, cw, boxes, C, dtype_code(net.dtype), _stream())


def leaky_bwd(A, dA, dZ, n, alpha):
    call('yolo2_leaky_bwd', ptr(A), ptr(dA), ptr(dZ), n, alpha, dtype_code(A.dtype), _stream())


def dropout(X, Y, mask, n, keep_prob, seed):
    call('yolo2_dropout', ptr(X), ptr(Y), ptr(mask), n, keep_prob, int(seed), dtype_code(X.dtype), _stream())


def dropout_bwd(dY, mask, dX, n, keep_prob):
    call('yolo2_dropout_bwd', ptr(dY), ptr(mask), ptr(dX), n, keep_prob, dtype_code(dY.dtype), _stream())


def l2_regularizer(w, g, n, scale, loss):
    call('yolo2_l2_regularizer', ptr(w), ptr(g), n, scale, ptr(loss), _stream())


# ---- image layer fused with its consumers (include/yolo2_hip.h: yolo2_first_layer_*)
def first_layer_stats(P, F, B, H, W, shift, bn_part):
    call('yolo2_first_layer_stats', ptr(P), ptr(F), B, H, W, ptr(shift), ptr(bn_part), dtype_code(P.dtype), _stream())


def first_layer_bn_leaky_pool(P, F, mean, var, gamma, beta, Pout, idx, B, H, W, ldp, eps, alpha):
    call('yolo2_first_layer_bn_leaky_pool', ptr(P), ptr(F), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(Pout), ptr(idx), B, H, W, ldp, eps, alpha,
         dtype_code(P.dtype), _stream())


def first_layer_pool_bwd_reduce(P, F, dP, lddp, idx, mean, var, gamma, beta, bn_part, B, H, W, eps, alpha):
    call('yolo2_first_layer_pool_bwd_reduce', ptr(P), ptr(F), ptr(dP), lddp, ptr(idx), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(bn_part), B, H, W,
         eps, alpha, dtype_code(P.dtype), _stream())


def first_layer_pool_bwd_apply(P, F, dP, lddp, idx, mean, var, gamma, beta, dgamma, dbeta, dY, B, H, W, eps, alpha):
    call('yolo2_first_layer_pool_bwd_apply', ptr(P), ptr(F), ptr(dP), lddp, ptr(idx), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(dgamma), ptr(dbeta),
         ptr(dY), B, H, W, eps, alpha, dtype_code(P.dtype), _stream())


# ---- data-parallel support
def set_stream_workgroups(n):
    """Stream-K convolution launches use ``n`` workgroups instead of one per CU (0 = default)."""
    call('yolo2_set_stream_workgroups', int(n))


def get_stream_workgroups():
    return int(_lib.load().yolo2_get_stream_workgroups())


def cast_f32_bf16(src, dst, n):
    call('yolo2_cast_f32_bf16', ptr(src), ptr(dst), n, _stream())


def cast_bf16_f32(src, dst, n):
    call('yolo2_cast_bf16_f32', ptr(src), ptr(dst), n, _stream())


def debug_occupy(workgroups, stop, started, max_us):
    call('yolo2_debug_occupy', workgroups, ptr(stop), ptr(started), max_us, _stream())

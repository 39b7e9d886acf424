"""ctypes binding of include/yolo2_hip.h (libyolo2hip.so, hand-written HIP for gfx950).

There is NO fallback: if the shared library is missing or a symbol cannot be resolved, importing
the compute path raises.  Every wrapper converts a non-zero status into ``HipKernelError`` carrying
``yolo2_last_error()``, mirroring the reference's behaviour of raising Python exceptions
(assert / tf.check_numerics) instead of returning codes.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('YOLO2_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libyolo2hip.so')     # the override is for timing-ablation builds (scripts/abl_build.sh)

F32, BF16 = 0, 1


class HipKernelError(RuntimeError):
    pass


_p = ctypes.c_void_p
_i = ctypes.c_int
_l = ctypes.c_long
_f = ctypes.c_float

# name -> argument ctypes (return type is always int unless noted)
SIGNATURES = {
    'yolo2_conv2d': [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    'yolo2_conv2d_ws': [_p, _p, _p, _p, _p, ctypes.c_size_t, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    'yolo2_conv2d_bias_leaky': [_p, _p, _p, _p, _p, ctypes.c_size_t, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _p],
    'yolo2_conv2d_dgrad_bn': [_p, _p, _p, _p, ctypes.c_size_t, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _f, _p, _i, _p],
    'yolo2_bn_part_to_grads': [_p, _i, _p, _p, _p],
    'yolo2_debug_noop': [_p],
    'yolo2_conv2d_wgrad': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    'yolo2_filter_prep': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'yolo2_filter_prep_batch': [_p, _i, _i, _i, _p],
    'yolo2_adam_filter_prep': [_p, _i, _i, _p, _i, _p, _p, _p, _p, _f, _f, _f, _f, _f, _i, _p],
    'yolo2_first_layer_stats': [_p, _p, _i, _i, _i, _p, _p, _i, _p],
    'yolo2_first_layer_bn_leaky_pool': [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _i, _p],
    'yolo2_first_layer_pool_bwd_reduce': [_p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _f, _i, _p],
    'yolo2_first_layer_pool_bwd_apply': [_p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _f, _i, _p],
    'yolo2_conv2d_bn': [_p, _p, _p, _p, ctypes.c_size_t, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _i, _p],
    'yolo2_bn_finalize': [_p, _p, _l, _i, _p, _p, _p, _p, ctypes.c_double, _p],
    'yolo2_bn_leaky_fin': [_p, _p, _i, _p, _p, _p, _p, _p, ctypes.c_double, _p, _p, _p, _l, _i, _i, _f, _f, _p, _l, _i, _p],
    'yolo2_bn_leaky_pool_fin': [_p, _p, _i, _p, _p, _p, _p, _p, ctypes.c_double, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _p, _l, _i, _p],
    'yolo2_bn_leaky_bwd_apply_fin': [_p, _i, _p, _p, _p, _p, _p, _p, _i, _l, _p, _p, _p, _l, _i, _f, _f, _p, _l, _i, _p],
    'yolo2_first_layer_wgrad_bn': [_p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _l, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p, _l, _i, _p],
    'yolo2_bn_leaky_pool_bwd_apply_fin': [_p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _l, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p, _l, _i, _p],
    'yolo2_bn_leaky_bwd_reduce_part': [_p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _l, _i, _f, _f, _i, _p],
    'yolo2_bn_leaky_pool_bwd_reduce_part': [_p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _i, _p],
    'yolo2_bn_leaky_pool': [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _i, _p],
    'yolo2_bn_leaky_pool_bwd_reduce': [_p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _i, _p],
    'yolo2_bn_leaky_pool_bwd_apply': [_p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _i, _p],
    'yolo2_augment_images': [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    'yolo2_transform_labels': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p],
    'yolo2_bn_stats': [_p, _p, _p, _p, _l, _i, _i, _p],
    'yolo2_bn_stats_ema': [_p, _p, _p, _p, _p, ctypes.c_double, _p, _l, _i, _i, _p],
    'yolo2_bn_ema': [_p, _p, _p, _p, _i, ctypes.c_double, _p],
    'yolo2_bn_leaky': [_p, _p, _p, _p, _p, _p, _l, _i, _i, _f, _f, _i, _p],
    'yolo2_bn_leaky_bwd_reduce': [_p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _l, _i, _f, _f, _i, _p],
    'yolo2_bn_leaky_bwd_apply': [_p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _l, _i, _f, _f, _i, _p],
    'yolo2_maxpool_fwd': [_p, _p, _i, _i, _i, _i, _i, _i, _p],
    'yolo2_maxpool_bwd': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'yolo2_maxpool_bwd_acc': [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    'yolo2_reorg': [_p, _p, _i, _i, _i, _i, _i, _i, _p],
    'yolo2_reorg_bwd': [_p, _i, _p, _i, _i, _i, _i, _i, _p],
    'yolo2_copy_channels': [_p, _i, _p, _i, _l, _i, _i, _p],
    'yolo2_add_inplace': [_p, _p, _l, _i, _p],
    'yolo2_bias_grad': [_p, _i, _p, _p, _l, _i, _i, _p],
    'yolo2_image_prep': [_p, _p, _p, _i, _i, _i, _i, _p],
    'yolo2_head_decode': [_p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'yolo2_head_decode_attrs': [_p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'yolo2_loss': [_p, _i, _p, _p, _p, _p, _p, _p, _p, ctypes.POINTER(_f), _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'yolo2_loss_partials': [_p, _i, _p, _p, _p, _p, _p, _p, _p, ctypes.POINTER(_f), _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'yolo2_loss_objectives': [_p, _p, _i, _i, _i, _i, _p],
    'yolo1_loss': [_p, _i, _p, _p, _p, _p, _p, _p, ctypes.POINTER(_f), _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'yolo1_head_decode': [_p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'yolo2_leaky_bwd': [_p, _p, _p, _l, _f, _i, _p],
    'yolo2_dropout': [_p, _p, _p, _l, _f, ctypes.c_ulonglong, _i, _p],
    'yolo2_dropout_bwd': [_p, _p, _p, _l, _f, _i, _p],
    'yolo2_l2_regularizer': [_p, _p, _l, _f, _p, _p],
    'yolo2_nms': [_p, _p, _p, _p, _p, _i, _i, _i, _f, _f, _p],
    'yolo2_adam': [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _p],
    'yolo2_momentum': [_p, _p, _p, _l, _f, _f, _f, _p],
    'yolo2_sgd': [_p, _p, _l, _f, _f, _p],
    'yolo2_rmsprop': [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _p],
    'yolo2_adagrad': [_p, _p, _p, _l, _f, _f, _p],
    'yolo2_adadelta': [_p, _p, _p, _p, _l, _f, _f, _f, _f, _p],
    'yolo2_clip_by_norm': [_p, _p, _i, _f, _p, _p],
    'yolo2_ftrl': [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _p],
    'yolo2_scale': [_p, _l, _f, _p],
    'yolo2_zero_ranges': [_p, _p, _i, _p],
    'yolo2_bn_fold': [_p, _p, _p, _p, _p, _p, _p, _l, _i, _f, _p],
    'yolo2_selftest_tr16': [_p, _p],
    'yolo2_set_stream_workgroups': [_i],
    'yolo2_cast_f32_bf16': [_p, _p, _l, _p],
    'yolo2_cast_bf16_f32': [_p, _p, _l, _p],
    'yolo2_debug_occupy': [_i, _p, _p, _i, _p],
    'yolo2_check_async_errors': [_p],
    'yolo2_async_error_snapshot': [_p, _p],
    'yolo2_debug_set_streamk_wait_us': [_i, _i],
}

# host queries / diagnostics: (restype, argtypes); bound in load() next to the status-returning entries above
QUERIES = {
    'yolo2_abi_version': (_i, []),
    'yolo2_last_error': (ctypes.c_char_p, []),
    'yolo2_shutdown': (_i, []),
    'yolo2_crc32c': (ctypes.c_uint32, [_p, ctypes.c_size_t, ctypes.c_uint32]),
    'yolo2_conv2d_wgrad_accumulates': (_i, [_i] * 9),            # 0 / 1, not a status
    'yolo2_filter_prep_blocks': (_i, [_i, _i, _i]),               # a count, not a status
    'yolo2_conv2d_dgrad_bn_fuses': (_i, [_i] * 6),               # 0 / 1, not a status
    'yolo2_debug_set_wgrad_variant': (None, [_i]),
    'yolo2_last_bn_part_rows': (_i, []),
    'yolo2_get_stream_workgroups': (_i, []),
    'yolo2_bn_fin_supported': (_i, [_i, _i, _i]),
    'yolo2_debug_last_conv_plan': (_i, [ctypes.POINTER(_i)]),
    'yolo2_debug_set_igemm_tap': (_i, [_i]),
    'yolo2_debug_set_pp': (_i, [_i, _i, _i, _i]),
    'yolo2_debug_set_pp_cost': (_i, [_i]),
    'yolo2_debug_set_s4_abl': (_i, [_i]),
    'yolo2_debug_last_wgrad_plan': (_i, [ctypes.POINTER(_i)]),
    'yolo2_debug_wgrad_row_plan': (_i, [_i, _i, _i, _i, _i, _i, _i, ctypes.POINTER(_i)]),
    'yolo2_debug_magic_u32': (_i, [ctypes.c_uint, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint)]),
    'yolo2_conv2d_workspace_bytes': (ctypes.c_size_t, [_i] * 7),
    'yolo2_bn_workspace_bytes': (ctypes.c_size_t, [_i]),
    'yolo2_bias_grad_workspace_bytes': (ctypes.c_size_t, [_i]),
    'yolo2_image_prep_workspace_bytes': (ctypes.c_size_t, [_i]),
    'yolo2_loss_workspace_bytes': (ctypes.c_size_t, [_i, _i, _i]),
    'yolo2_nms_workspace_bytes': (ctypes.c_size_t, [_i, _i, _i]),
    'yolo2_clip_workspace_bytes': (ctypes.c_size_t, [_i]),
    'yolo2_augment_workspace_bytes': (ctypes.c_size_t, [_i]),
}


class AugmentParams(ctypes.Structure):
    """yolo2_augment_params (include/yolo2_hip.h)."""
    _fields_ = [('src_offset', ctypes.c_longlong), ('src_w', _i), ('src_h', _i), ('crop_x', _i), ('crop_y', _i), ('crop_w', _i), ('crop_h', _i),
                ('flags', ctypes.c_uint), ('brightness', _f), ('saturation', _f), ('hue', _f), ('contrast', _f), ('noise_scale', _f),
                ('noise_seed', ctypes.c_ulonglong)]


class FilterDesc(ctypes.Structure):
    """yolo2_filter_desc of include/yolo2_hip.h"""
    _fields_ = [('W', ctypes.c_void_p), ('Ffwd', ctypes.c_void_p), ('Fdgr', ctypes.c_void_p), ('ksize', _i), ('cin', _i),
                ('ldcin', _i), ('cout', _i), ('ldcout', _i), ('first_block', _i)]


_lib = None
MISSING = set()      # entry points a baseline library (YOLO2_LIB_BASELINE=1) does not export


def _mapped_hip_runtimes():
    try:
        with open('/proc/self/maps') as f:
            return sorted({line.split()[-1] for line in f if 'libamdhip64' in line})
    except OSError:
        return []


def load():
    """Loads libyolo2hip.so and resolves every symbol of the header; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipKernelError(
            'HIP extension %s is missing: run `python yolo_tf_amd/csrc/build.py` (or __graft_entry__.build()). '
            'There is no CPU fallback.' % LIB_PATH)
    # torch FIRST: its wheel bundles its own libamdhip64.so (same SONAME as /opt/rocm's).  If this library were loaded before
    # torch, it would bind /opt/rocm's copy and torch would then map a SECOND HIP runtime: two null streams with no ordering
    # between torch's copies and these kernels, and device pointers one runtime allocated are unknown to the other's checked
    # calls (hipMemsetAsync fails with "invalid value").  Loaded after torch, the dynamic linker reuses the runtime already mapped.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    runtimes = _mapped_hip_runtimes()
    if len(runtimes) > 1:
        raise HipKernelError('two HIP runtimes are mapped in this process (%s): something loaded %s before torch' % (', '.join(runtimes), LIB_PATH))
    # Same-box A/B against an EARLIER round's library (profiles/baseline/libyolo2hip_rNN.so through YOLO2_LIB_PATH, scripts/gpu_evidence.sh
    # section `rounds`): YOLO2_LIB_BASELINE=1 tolerates entry points that library predates -- calling one raises, nothing falls back.
    baseline = os.environ.get('YOLO2_LIB_BASELINE') == '1' and bool(os.environ.get('YOLO2_LIB_PATH'))

    def resolve(name):
        try:
            return getattr(lib, name)       # AttributeError if the symbol is not exported
        except AttributeError:
            if not baseline:
                raise
            MISSING.add(name)

            def absent(*_args, _name=name):
                raise HipKernelError('%s is not exported by the baseline library %s' % (_name, LIB_PATH))
            setattr(lib, name, absent)
            return None
    for name, (res, args) in QUERIES.items():
        fn = resolve(name)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    for name, args in SIGNATURES.items():
        fn = resolve(name)
        if fn is not None:
            fn.restype = _i
            fn.argtypes = args
    _lib = lib
    over = env_overrides()
    if over:      # A/B switches select kernel variants: say so once, so that a run that differs from the tested defaults is not silent about it
        import warnings
        warnings.warn('YOLO2_* switches set in the environment change kernel selection away from the tested defaults: %s' % ', '.join(over))
    return lib


def env_overrides():
    """The YOLO2_* variables present in the environment (library and engine A/B switches), as 'NAME=value' strings.
    Everything the parity tests and the driver's benchmark cover is the state in which this list is empty."""
    return sorted('%s=%s' % (k, v) for k, v in os.environ.items() if k.startswith('YOLO2_'))


def query(name, *args):
    """Host-side query of include/yolo2_hip.h (workspace sizes, plans): returns the raw value."""
    return getattr(load(), name)(*args)


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise HipKernelError('%s failed (code %d): %s' % (name, rc, lib.yolo2_last_error().decode()))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def dtype_code(torch_dtype):
    import torch
    if torch_dtype == torch.float32:
        return F32
    if torch_dtype == torch.bfloat16:
        return BF16
    raise ValueError('unsupported dtype %r' % (torch_dtype,))

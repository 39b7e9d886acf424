"""CPU-side boundary checks: the C-ABI library builds for gfx950, loads, and exports every symbol
include/yolo2_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'yolo2_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(yolo[12]_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    import torch  # noqa: F401  (one HIP runtime per process: torch's, see yolo_tf_amd/_lib.py load())
    from yolo_tf_amd.csrc import build
    path = build.build(verbose=False)
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 28
    for n in names:
        assert hasattr(lib, n), 'missing symbol %s' % n
    lib.yolo2_abi_version.restype = ctypes.c_int
    assert lib.yolo2_abi_version() == 1


def test_binding_table_matches_header():
    from yolo_tf_amd import _lib
    declared = set(_declared())
    bound = set(_lib.SIGNATURES) | set(_lib.QUERIES)          # status-returning entries | host queries and diagnostics
    assert declared == bound, declared ^ bound
    assert not set(_lib.SIGNATURES) & set(_lib.QUERIES)
    _lib.load()


def test_workspace_queries_are_host_only_and_match_the_documented_sizes():
    from yolo_tf_amd import _lib
    q = _lib.query
    assert q('yolo2_bn_workspace_bytes', 1024) == 1025 * 1024 * 8
    assert q('yolo2_bias_grad_workspace_bytes', 432) == 512 * 432 * 8
    assert q('yolo2_image_prep_workspace_bytes', 16) == 2 * 64 * 16 * 8        # 64 partial (sum, sum of squares) pairs per image
    assert q('yolo2_nms_workspace_bytes', 256, 845, 20) == 256 * 845 * 20 * 4
    assert q('yolo2_loss_workspace_bytes', 16, 169, 5) == (4 * ((16 * 169 * 8 + 255) // 256) + 4) * 4
    assert q('yolo2_clip_workspace_bytes', 66) == 66 * 8 and q('yolo2_augment_workspace_bytes', 16) == 3 * 16 * 8
    assert q('yolo2_shutdown') == 0                      # nothing allocated yet: a no-op


def test_argument_errors_raise_without_touching_the_gpu():
    import pytest
    from yolo_tf_amd import _lib
    with pytest.raises(_lib.HipKernelError, match='argument check failed'):
        _lib.call('yolo2_conv2d', None, None, None, None, 1, 1, 1, 8, 8, 8, 8, 3, 0, None)
    with pytest.raises(_lib.HipKernelError):
        _lib.call('yolo2_nms', None, None, None, None, None, 1, 10, 2, 0.3, 0.4, None)


def test_single_hip_runtime_after_load():
    """Loading the C-ABI library must not map a second libamdhip64 next to torch's bundled one (stream ordering and pointer
    validation both break across runtimes); _lib.load() imports torch first and checks."""
    from yolo_tf_amd import _lib
    _lib.load()
    assert len(_lib._mapped_hip_runtimes()) == 1, _lib._mapped_hip_runtimes()


def test_missing_library_fails_loudly(monkeypatch):
    import pytest
    from yolo_tf_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libyolo2hip.so')
    with pytest.raises(_lib.HipKernelError, match='no CPU fallback'):
        _lib.load()

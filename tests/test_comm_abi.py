"""include/yolo2_comm.h <-> libyolo2comm.so (the RCCL communicator lifecycle + bucket all-reduce for C / C++ hosts).
CPU part: the library loads in a process WITHOUT torch (it links /opt/rocm's RCCL; the Python host never maps it), exports exactly
the header's functions, and reports argument / no-device errors as codes + messages.  GPU part (tests/test_comm_gpu.py) builds the
C++ host of examples/dp_host.cpp and runs it."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'yolo2_comm.h')
LIB = os.path.join(ROOT, 'yolo_tf_amd', 'csrc', 'libyolo2comm.so')


def _declared():
    text = re.sub(r'/\*.*?\*/', '', open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r'\b(yolo2_comm_\w+)\s*\(', text)))


def test_header_and_library_export_the_same_functions():
    from yolo_tf_amd.csrc import build
    build.build_comm(verbose=False)
    out = subprocess.check_output(['nm', '-D', '--defined-only', LIB], text=True)
    exported = sorted(set(re.findall(r' T (yolo2_comm_\w+)', out)))
    assert exported == _declared(), (exported, _declared())
    assert {'yolo2_comm_init', 'yolo2_comm_allreduce_bucket', 'yolo2_comm_destroy'} <= set(exported)      # SURVEY 8b's three


def test_loads_without_torch_and_reports_errors_as_codes():
    code = r'''
import ctypes, sys
assert 'torch' not in sys.modules
lib = ctypes.CDLL(sys.argv[1])
lib.yolo2_comm_last_error.restype = ctypes.c_char_p
assert lib.yolo2_comm_allreduce_bucket(None, None, ctypes.c_long(4), 0, None) == 1            # YOLO2_COMM_E_ARG
assert b'allreduce_bucket' in lib.yolo2_comm_last_error()
assert lib.yolo2_comm_broadcast(None, None, ctypes.c_long(4), 0, None) == 1
comm = ctypes.c_void_p()
assert lib.yolo2_comm_init(ctypes.byref(comm), None, 0, 1, 0) == 1 and not comm.value
assert lib.yolo2_comm_init(ctypes.byref(comm), (ctypes.c_char * 128)(), 3, 2, 0) == 1          # rank outside the world
assert lib.yolo2_comm_destroy(None) == 0 and lib.yolo2_comm_rank(None) == -1 and lib.yolo2_comm_world(None) == -1
print('ok')
'''
    out = subprocess.check_output([sys.executable, '-c', code, LIB], text=True, stderr=subprocess.STDOUT)
    assert out.strip().endswith('ok'), out


def test_python_host_does_not_map_the_comm_library():
    """One RCCL per process: the torch.distributed host must never load libyolo2comm.so (it would bring /opt/rocm's RCCL + HIP runtime
    next to torch's own).  _lib.SIGNATURES is the complete binding table of the Python host."""
    from yolo_tf_amd import _lib
    assert not any(name.startswith('yolo2_comm_') for name in _lib.SIGNATURES)
    src = open(os.path.join(ROOT, 'yolo_tf_amd', '_lib.py')).read() + open(os.path.join(ROOT, 'yolo_tf_amd', 'parallel.py')).read()
    assert 'libyolo2comm' not in src

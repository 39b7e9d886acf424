"""Builds libyolo2hip.so (the C-ABI library of include/yolo2_hip.h) for gfx950 with hipcc.

In-tree, explicit: `python yolo_tf_amd/csrc/build.py` (also called by __graft_entry__.build()).
hipcc cross-compiles without a GPU.  The exact-compare kernels (loss IoU equality, NMS) are built
with FP contraction off."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = {
    'conv_igemm.hip': [],
    'conv_pp.hip': [],
    'conv_s4.hip': [],
    'conv_wgrad.hip': [],
    'conv_wgrad3.hip': [],
    'conv_wgrad_c32.hip': [],
    'conv_c32.hip': [],
    'conv_c64.hip': [],
    'conv_d1.hip': [],
    'conv_first.hip': [],
    'elementwise.hip': [],
    'head.hip': ['-ffp-contract=off'],
    'yolo1.hip': ['-ffp-contract=off'],
    'nms.hip': ['-ffp-contract=off'],
    'augment.hip': ['-ffp-contract=off'],
}
COMMON = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-Wno-unused-result']
LIB = os.path.join(HERE, 'libyolo2hip.so')
COMM_LIB = os.path.join(HERE, 'libyolo2comm.so')     # include/yolo2_comm.h: RCCL communicator + bucket all-reduce for C / C++ hosts
ROCM = os.environ.get('ROCM_PATH', '/opt/rocm')


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', os.path.join(ROCM, 'bin', 'hipcc'))
    headers = [os.path.join(HERE, 'common.h'), os.path.join(HERE, 'conv_shared.h'), os.path.join(HERE, '..', '..', 'include', 'yolo2_hip.h'), os.path.abspath(__file__)]
    objs, cmds = [], []
    for src, extra in SOURCES.items():
        s = os.path.join(HERE, src)
        o = os.path.join(HERE, src.replace('.hip', '.o'))
        if force or _stale(o, [s] + headers):
            cmds.append([hipcc] + COMMON + extra + ['-c', s, '-o', o])
        objs.append(o)
    if cmds:      # independent translation units: compile them side by side (YOLO2_BUILD_JOBS, default: the CPUs of this host, at most 8)
        from concurrent.futures import ThreadPoolExecutor
        jobs = max(1, min(int(os.environ.get('YOLO2_BUILD_JOBS', min(8, os.cpu_count() or 1))), len(cmds)))
        if verbose:
            for cmd in cmds:
                print(' '.join(cmd), flush=True)
        with ThreadPoolExecutor(jobs) as pool:
            list(pool.map(subprocess.check_call, cmds))
    if force or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        # fails here (not on the GPU box) if any symbol is unresolved.  In a child process: dlopen()ing the library in THIS
        # process before torch is imported would map a second HIP runtime next to torch's bundled one (see _lib.load)
        subprocess.check_call([sys.executable, '-c', 'import ctypes, sys; ctypes.CDLL(sys.argv[1])', LIB])
    # the communicator library is optional: a ROCm install without RCCL development files still gets the single-GPU library (the Python
    # host uses torch's own RCCL; only C / C++ hosts that want data parallelism need libyolo2comm.so)
    if os.path.exists(os.path.join(ROCM, 'include', 'rccl', 'rccl.h')) or os.path.exists(os.path.join(ROCM, 'include', 'rccl.h')):
        try:
            build_comm(force=force, verbose=verbose)
        except subprocess.CalledProcessError as exc:
            print('warning: libyolo2comm.so not built (%s); include/yolo2_comm.h hosts are unavailable' % exc, file=sys.stderr)
    elif verbose:
        print('note: no rccl.h under %s: libyolo2comm.so skipped' % ROCM, file=sys.stderr)
    return LIB


def build_comm(force=False, verbose=True):
    """libyolo2comm.so: host code only (RCCL launches its own kernels).  A separate shared object on purpose: it links /opt/rocm's
    librccl, and the Python host (torch.distributed, torch's own RCCL) must not map it -- see include/yolo2_comm.h."""
    hipcc = os.environ.get('HIPCC', os.path.join(ROCM, 'bin', 'hipcc'))
    src = os.path.join(HERE, 'comm.cpp')
    hdr = os.path.join(HERE, '..', '..', 'include', 'yolo2_comm.h')
    if force or _stale(COMM_LIB, [src, hdr, os.path.abspath(__file__)]):
        cmd = [hipcc, '-O2', '-std=c++17', '-fPIC', '-shared', '-o', COMM_LIB, src, '-I' + os.path.join(ROCM, 'include'),
               '-L' + os.path.join(ROCM, 'lib'), '-lrccl', '-lamdhip64', '-Wl,-rpath,' + os.path.join(ROCM, 'lib')]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        subprocess.check_call([sys.executable, '-c', 'import ctypes, sys; ctypes.CDLL(sys.argv[1])', COMM_LIB])
    return COMM_LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)

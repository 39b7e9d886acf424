"""A C++ host (examples/dp_host.cpp) drives the data-parallel exchange through the two C ABIs alone -- libyolo2comm.so (RCCL communicator,
bucket all-reduce, broadcast, agree) and libyolo2hip.so (bf16 wire casts, the optimizer kernel) -- one process per GPU, no Python in
the ranks.  Runs with one rank on the test box and with one rank per device when it has more."""
import os
import subprocess

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'yolo_tf_amd', 'csrc')


@pytest.fixture(scope='module')
def dp_host(tmp_path_factory):
    from yolo_tf_amd.csrc import build
    build.build(verbose=False)
    exe = str(tmp_path_factory.mktemp('dp') / 'dp_host')
    subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '-O2', '-std=c++17', os.path.join(ROOT, 'examples', 'dp_host.cpp'),
                           '-I' + os.path.join(ROOT, 'include'), '-L' + CSRC, '-lyolo2hip', '-lyolo2comm', '-Wl,-rpath,' + CSRC, '-o', exe])
    return exe


@pytest.mark.parametrize('world', [1, 2, 8])
def test_cpp_host_data_parallel_exchange(dp_host, world):
    if world > torch.cuda.device_count():
        pytest.skip('%d GPUs needed, %d visible' % (world, torch.cuda.device_count()))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([dp_host, str(world)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert r.stdout.count('OK') == world, r.stdout

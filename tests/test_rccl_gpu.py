"""RCCL is loaded and driven by the production data-parallel code at least once on the 1-GPU test box: a one-rank "nccl"
process group (backend nccl = RCCL on ROCm) runs GradReducer's bucketed all-reduce on its communication stream, a whole
training step whose buckets are consumed by the optimizer as they complete, and the start-up broadcast.  The N > 1 semantics
(sums over ranks, replicas staying identical) are covered with gloo in tests/test_parallel_cpu.py and
test_network_gpu.py::test_data_parallel_step_two_processes_one_gpu; 8-GPU runs belong to the driver."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_rccl_training_step():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', '_rccl_worker.py')], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'RCCL_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]

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


@pytest.mark.parametrize('extra', [['--multiscale'], ['--shard-optimizer']], ids=['multiscale', 'sharded-optimizer'])
def test_bench_two_ranks_prints_its_json_line(extra):
    """bench.py's N > 1 legs end to end (BASELINE configs[3] is a multi-GPU multi-scale run: its post-timing communication report once
    passed the size->images dict to step() and died before the JSON line): two ranks under torch.distributed.run share the one GPU of the
    test box (gloo; the RCCL path itself is the one-rank test above), a few steps, and rank 0 prints ONE line with the whole-job fields."""
    import json
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--batch', '2', '--backend', 'gloo', '--no-cpu-baseline',
           '--no-detect', '--no-kernel-timer'] + extra
    r = subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'), capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-4000:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['steps'] == 3 and j['value'] > 0 and j['scaling'] == 'weak'
    assert j['comm'] and len(j['comm']['buckets']) >= 1, j.get('comm')

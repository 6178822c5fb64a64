"""The sharded LM loop on REAL RCCL with more than one rank: two (or up to
four) fresh child processes, one GPU each, started BEFORE this process touches a GPU for them (torch.distributed.run spawns them;
nothing is exec'd from a process that has initialised HIP).  Skipped on boxes with a single GPU -- the ranks-as-threads tests of
tests/test_dist_c_gpu.py cover the loop's logic there; this one covers ncclCommInitRank per process, asynchronous collectives
behind the epoch gate, and the final N x 10 all-reduce."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('world', [2, 4])
def test_sharded_loop_over_rccl_matches_the_fused_loop(world):
    if torch.cuda.device_count() < world:           # (device_count() does not initialise HIP on this image)
        pytest.skip('needs %d GPUs, this box has %d' % (world, torch.cuda.device_count()))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONPATH=ROOT)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', 'rccl_worker.py')]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert 'failed_solve ok' in out.stdout and 'noisy65a ok' in out.stdout

"""World-size-2 run of the sharded path on the GPU box: every page's ids equal the single-process result.
On a single-GPU box both ranks share GPU 0 and the all-gather goes over gloo (host buffers); on a multi-GPU node set
CR_DIST_BACKEND=nccl to exercise RCCL."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_pages_equal_single_process():
    env = dict(os.environ)
    env.setdefault('CR_DIST_BACKEND', 'nccl' if torch.cuda.device_count() >= 2 else 'gloo')
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29533', os.path.join(ROOT, 'scripts', 'dist_check.py')]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    assert 'DIST_CHECK OK' in out, out[-3000:]


def test_rccl_all_gather_runs_on_the_visible_gpus():
    """backend 'nccl' = RCCL.  With >= 2 GPUs: two ranks, one GPU each.  On a one-GPU box: world size 1 -- communicator set-up and
    the collective kernels still run (RCCL cannot put two ranks on one GPU, so the world-2 equality test above uses gloo there)."""
    n = min(torch.cuda.device_count(), 2)
    env = dict(os.environ)
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    script = os.path.join(ROOT, 'scripts', 'rccl_check.py')
    if n >= 2:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
               '--master-port', '29537', script]
    else:
        cmd = [sys.executable, script]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and 'RCCL_CHECK OK' in out, out[-3000:]

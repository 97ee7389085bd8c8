"""GPU: the data-parallel PRODUCT step (TrainStep with parallel.GradSync over RCCL) against the single-GPU step —
SURVEY §8e. Launched exactly like the benchmark: one fresh child process per rank through torch.distributed.run on
127.0.0.1 (the RCCL process group must not live inside the pytest process: its watchdog thread breaks hipGraph capture
in the other tests). World size 1 runs on any box; world size 2 needs two visible GPUs and skips otherwise.
The worker (tests/dp_gpu_worker.py) checks: losses, parameters and parameter updates after 3 iterations equal the
single-GPU step on the global batch; gradient segments are handed to RCCL while the backward pass is still running;
capture() refuses data-parallel steps; all ranks end bit-identical."""
import os
import subprocess
import sys

import pytest
import torch

from common import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('world', [1, 2])
def test_dp_train_step_rccl(world, dev):
    if torch.cuda.device_count() < world:
        pytest.skip(f'{world} GPUs needed, {torch.cuda.device_count()} visible')
    port = 29600 + (os.getpid() + world) % 2000
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tests', 'dp_gpu_worker.py')]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='4')
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert 'DP_GPU_OK' in r.stdout, tail
    print(r.stdout.strip().splitlines()[-1])

"""CPU, world_size 2, gloo: the data-parallel gradient path (parallel.GradSync over a flat gradient arena) and
the sharding contract of SURVEY §8e — with equal shards, the mean of the rank gradients equals the gradient of
the global batch (checked with the CPU oracle's discriminator step). Launched exactly like the benchmark:
one process per rank through torch.distributed.run on 127.0.0.1."""
import os
import subprocess
import sys

from common import ROOT


def test_gradsync_two_ranks_gloo():
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tests', 'dp_worker.py')]
    env = dict(os.environ, OMP_NUM_THREADS='2')
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'DP_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]

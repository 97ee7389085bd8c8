import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def pkg():
    return importlib.import_module('td-vc-gan_amd')


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


def rel_l2(a, b):
    import torch
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def pytest_sessionfinish(session, exitstatus):
    """Outlier counts of every gradient gate of the session (tests/common.assert_grads_close) -> gpurun_out/grad_outliers.json."""
    import json
    common = sys.modules.get('common')
    log = getattr(common, 'GRAD_OUTLIER_LOG', None) if common else None
    if log:
        out = os.path.join(ROOT, 'gpurun_out')
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'grad_outliers.json'), 'w') as f:
            json.dump(log, f, indent=1)

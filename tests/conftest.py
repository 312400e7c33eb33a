import os
import sys

import pytest

# golden traces and multi-rank trajectories are those of the in-tree samplers: a machine that has
# pycma / nevergrad installed must not silently run the tests on other sampler numerics
# (tests/test_backends.py exercises the external branch with recording fakes)
os.environ.setdefault('P2L_SAMPLERS', 'intree')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    # the GPU box has 256 host cores; torch-CPU (the oracle) is slowest when it
    # oversubscribes them
    import torch
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda')

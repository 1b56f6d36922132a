import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def c_oracle():
    from oracle import c_oracle as co
    co.load()
    return co


@pytest.fixture(scope='session')
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail('this test needs a HIP device; the path has no CPU fallback')
    return torch.device('cuda:0')


@pytest.fixture
def tuning():
    """yh_set_tuning for the duration of a test; the defaults come back afterwards."""
    from yolo_tf_amd import _abi
    yield _abi.set_tuning
    _abi.set_tuning()

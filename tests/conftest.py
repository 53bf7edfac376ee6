import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: longer statistical test")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (there is no CPU fallback)")
    return torch.device("cuda:0")

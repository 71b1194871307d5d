import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from lap_amd import hip as _hip  # raises if liblap_hip.so is missing: no fallback

    return _hip

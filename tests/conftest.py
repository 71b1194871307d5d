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


@pytest.fixture(autouse=True)
def _release_device_memory():
    """The full-depth tests hold 30 - 70 GB of LAP-3B state each; objects that died in reference cycles (autograd graphs of the oracle,
    the engine's saved contexts) are only freed by the cycle collector, and with several such tests in one process the 288 GB fill up
    (round 6: three full-depth training cases instead of one).  Collect and hand the cached blocks back after every test."""
    yield
    import gc

    gc.collect()
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.empty_cache()
    except Exception:  # noqa: BLE001
        pass

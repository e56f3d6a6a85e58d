import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no GPU is visible")
    import macvo_amd._lib as L

    L.load()  # fail loudly if libmacvo_hip.so is missing: GPU tests must exercise the HIP path
    return torch.device("cuda:0")

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionstart(session):
    """On a GPU box, let torch bring up ITS copy of the HIP runtime before libusher_amd.so (linked against /opt/rocm's)
    loads its own: the other order leaves torch with "No HIP GPUs are available" in tests that use both (bench.py imports
    torch first for the same reason).  device_count() itself does not touch the GPU."""
    try:
        import torch
        if torch.cuda.device_count() > 0:
            torch.cuda.init()
    except Exception:
        pass

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): oracle/mm_oracle.c through ctypes."""
    from oracle import mm_oracle
    mm_oracle.build()
    return mm_oracle


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library on a GPU box. Fails loudly (no fallback) if it cannot run."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import mmannot_amd
    mmannot_amd.lib()
    return mmannot_amd

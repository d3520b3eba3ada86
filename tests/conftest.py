import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box only)")


@pytest.fixture(scope="session")
def oracle_clib():
    """Builds the C part of the CPU oracle once (gcc, seconds)."""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    from oracle import dbscan
    dbscan._clib()
    return True


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test started without a GPU")
    from pointcloudhookup_amd import _lib
    _lib.lib()            # raises loudly when the extension is missing
    return torch.device("cuda:0")

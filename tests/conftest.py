import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the oracle (test infrastructure) and, when missing, the HIP library (hipcc cross-compiles on CPU)."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-j4"], check=True, capture_output=True)
    import delphy_amd
    if not os.path.exists(delphy_amd.library_path()):
        delphy_amd.build_library()
    yield

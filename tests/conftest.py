import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_count():
    """Number of HIP devices, asked of a child process so that collecting tests never initialises the GPU here."""
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
        return int(r.stdout.strip().splitlines()[-1])
    except Exception:
        return 0


_gpu_tests_will_run = False


def pytest_collection_modifyitems(config, items):
    global _gpu_tests_will_run
    gpu_items = [it for it in items if "gpu" in it.keywords and not any(m.name == "skip" for m in it.iter_markers())]
    selected = config.getoption("-m") or ""
    if gpu_items and _gpu_count() == 0:
        skip = pytest.mark.skip(reason="no HIP device: the engine has no CPU fallback (run with -m gpu on an MI355X)")
        for it in gpu_items:
            it.add_marker(skip)
    elif gpu_items and "not gpu" not in selected:
        _gpu_tests_will_run = True


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the oracle (test infrastructure) and, when missing, the HIP library (hipcc cross-compiles on CPU)."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-j4"], check=True, capture_output=True)
    import delphy_amd
    if not os.path.exists(delphy_amd.library_path()):
        delphy_amd.build_library()
    if _gpu_tests_will_run:
        # torch brings a HIP runtime of its own: it is initialised before the engine's library first touches the device, in whatever
        # order the tests run (the other way round, `torch.cuda` found no device in a process that had been using the engine for a while)
        import torch
        torch.cuda.init()
    yield

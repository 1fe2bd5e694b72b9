import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Every handle the GPU suite creates gets its workspace filled with NaN before the library initialises it (newton.py, batched.py,
# the library-owned allocations): a kernel that reads memory nobody wrote shows up instead of hiding behind the zero pages of a
# fresh process (round 4: a 128 x 256 update tile did exactly that).  Worker processes inherit it.
os.environ.setdefault("PYIPM_POISON_WORKSPACE", "1")
# (no global PYIPM_EXPERT: a test that drives an expert switch opens the gate on ITS handle with set_option("expert", 1) --
#  VERDICT r5 item 8; tests/test_gpu_symmetric.py checks the gate itself)
os.environ.pop("PYIPM_EXPERT", None)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

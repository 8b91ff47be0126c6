import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/): checker only, never the thing under test on -m gpu runs."""
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "host_matrix_kat.json")) as f:
        return json.load(f)


def _gpu_visible():
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        return False


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests need an MI355X: without one they are skipped, never silently passed."""
    if _gpu_visible():
        return
    skip = pytest.mark.skip(reason="no GPU visible (run through gpurun)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)

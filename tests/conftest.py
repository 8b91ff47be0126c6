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

import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "dct_golden.npz"))


@pytest.fixture(scope="session")
def scan_golden():
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "scan_golden.json")) as f:
        return json.load(f)

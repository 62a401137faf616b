import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "seamount_65x49x21.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_planes():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "seamount_65x49x21_planes.npz"))


@pytest.fixture(scope="session")
def golden_kb50():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "kb50_256x192x50.json")) as f:
        return json.load(f)

import json
import os
import subprocess
import sys
import tempfile
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver on the GPU box)")
    config.addinivalue_line("markers", "bg_oracle(case, im, jm, kb, steps): the CPU oracle's steps of this grid are computed by a child "
                                       "process (tests/oracle_bg.py) from the start of the session, beside the other tests")
    config._bg_oracle = {}


# ---- the oracle of the full-size grids, beside the other tests (tests/oracle_bg.py) ------------------------------------------------
class BgOracle:
    def __init__(self, key):
        self.key = key
        self.dir = tempfile.mkdtemp(prefix="pom_bg_oracle_")
        self.log = open(os.path.join(self.dir, "log.txt"), "w")
        self.proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "oracle_bg.py")] + [str(v) for v in key] + [self.dir],
                                     stdout=self.log, stderr=subprocess.STDOUT, cwd=ROOT)

    def step(self, n, timeout=900):
        """{array: digest} after oracle step n (0 = the initial state); waits for the child to get there"""
        path = os.path.join(self.dir, f"step{n}.json")
        t0 = time.time()
        while not os.path.exists(path):
            if self.proc.poll() is not None and not os.path.exists(path):
                raise RuntimeError(f"tests/oracle_bg.py {self.key} ended with {self.proc.returncode} before step {n}:\n" + open(self.log.name).read()[-3000:])
            if time.time() - t0 > timeout:
                raise RuntimeError(f"tests/oracle_bg.py {self.key}: step {n} not there after {timeout} s")
            time.sleep(0.5)
        with open(path) as f:
            return json.load(f)

    def close(self):
        if self.proc.poll() is None:
            self.proc.kill()                                  # exactly the child this session started
            self.proc.wait()
        self.log.close()
        import shutil
        shutil.rmtree(self.dir, ignore_errors=True)


def pytest_collection_finish(session):
    if session.config.option.collectonly:
        return
    for item in session.items:
        m = item.get_closest_marker("bg_oracle")
        if m and tuple(m.args) not in session.config._bg_oracle:
            session.config._bg_oracle[tuple(m.args)] = BgOracle(tuple(m.args))


def pytest_sessionfinish(session, exitstatus):
    for h in getattr(session.config, "_bg_oracle", {}).values():
        h.close()


@pytest.fixture
def bg_oracle(request):
    m = request.node.get_closest_marker("bg_oracle")
    return request.config._bg_oracle[tuple(m.args)]


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "seamount_65x49x21.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_planes():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "seamount_65x49x21_planes.npz"))


@pytest.fixture(scope="session")
def golden_kb50():
    with open(os.path.join(ROOT, "tests", "golden", "kb50_256x192x50.json")) as f:
        return json.load(f)

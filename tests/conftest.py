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
def pkg():
    from __graft_entry__ import load_package

    return load_package()


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    gdir = os.path.join(ROOT, "tests", "golden")

    def load(name):
        return np.load(os.path.join(gdir, name + ".npz"))

    return load

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

G2O_DIR = os.path.join(ROOT, "tests", "golden", "g2o")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def g2o_path(name):
    return os.path.join(G2O_DIR, name + ".g2o")


@pytest.fixture(scope="session")
def g2o():
    return g2o_path

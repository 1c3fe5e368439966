import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


@pytest.fixture(scope="session")
def pkg():
    graft.build()
    return graft.load_package()


@pytest.fixture(scope="session")
def orc():
    o = graft.load_oracle()
    o.build()
    return o

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as ge  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return ge.load_package()


@pytest.fixture(scope="session")
def oracle():
    o = ge.load_oracle()
    o.lib()  # builds on demand with gcc if the .so did not travel
    return o


@pytest.fixture(scope="session")
def engine(pkg):
    """one GPU context shared by the GPU tests (fails loudly if the HIP library or GPU is missing)"""
    eng = pkg.Engine(0)
    yield eng
    eng.close()

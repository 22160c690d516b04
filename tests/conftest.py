import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import checker
    return checker.load_oracle()


@pytest.fixture(scope="session")
def ref():
    import checker
    r = checker.load_ref()
    if r is None:
        pytest.skip("oracle/_ref/libaqref.so not built (reference tree not mounted)")
    return r

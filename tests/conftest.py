import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_IN = os.path.join(HERE, "golden", "ref_in")
GOLDEN_OUT = os.path.join(HERE, "golden", "ref_out")
RESOURCES = os.path.join(ROOT, "ngs-bits_amd", "resources")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gin():
    return lambda name: os.path.join(GOLDEN_IN, name)


@pytest.fixture(scope="session")
def gout():
    return lambda name: os.path.join(GOLDEN_OUT, name)

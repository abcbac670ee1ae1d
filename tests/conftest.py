import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` through gpurun)")


@pytest.fixture(scope="session")
def oracle_lib():
    import orc
    orc.build()
    return orc.lib()


@pytest.fixture(scope="session")
def engine():
    """A live engine on cuda:0 — fails (does not skip) when the HIP library or the GPU is missing."""
    import rvtests_amd
    eng = rvtests_amd.Engine(0)
    yield eng
    eng.close()

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout: pytest-timeout's marker (registered here too, so that a box without the plugin only ignores it)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/): the checker, never the thing under test on the product side."""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def oracle_backend(monkeypatch):
    """Route the host-side classes of mdapy_amd through the CPU oracle (test infrastructure).
    This exercises the Python policy layer (replication, list reuse, normalisation) on a machine
    without a GPU; the product package itself has no such switch."""
    import _oracle_backend as ob

    ob.install(monkeypatch)
    return ob

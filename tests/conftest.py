import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def built():
    """Make sure the native pieces exist (cross-compiles on CPU boxes)."""
    import __graft_entry__ as entry

    entry.build()
    return entry


def gpu_present():
    try:
        from pvtrace_amd.engine import native

        return native.is_available()
    except Exception:
        return False

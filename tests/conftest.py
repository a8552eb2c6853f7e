import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The HIP library must exist (tests never fall back to a CPU path)."""
    lib = os.path.join(ROOT, "yocto-gl_amd", "csrc", "libythip.so")
    if not os.path.exists(lib):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g
        g.build()
    yield

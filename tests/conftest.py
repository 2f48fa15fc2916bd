import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """The in-tree libraries (liblpipm.so + the oracle) -- built once per session if missing."""
    import __graft_entry__ as g
    from lp_amd import _capi
    if not os.path.exists(_capi.LIB_PATH) or not os.path.exists(os.path.join(ROOT, "oracle", "liboracle_ipm.so")):
        g.build()
    return True


@pytest.fixture(scope="session")
def ctx(built):
    import lp_amd
    return lp_amd.default_context(0)

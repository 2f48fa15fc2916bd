import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Environment knobs of liblpipm.so that CHANGE THE BITS of a result (summation chunking, super-block width, refinement,
# tile edges, schedules): parity is claimed for the library's defaults only, so none of them may leak in from the
# environment the suite runs in.  (Tests that exercise a knob set it themselves with monkeypatch, after this check.)
# The library ignores every one of them unless the master switch LPIPM_EXPERIMENTAL=1 is set (lp_knob, lpipm_internal.hpp).
BIT_CHANGING_KNOBS = ("LPIPM_EXPERIMENTAL", "LPIPM_ADAT_KC", "LPIPM_ADAT_SK", "LPIPM_SUPER", "LPIPM_REFINE", "LPIPM_REFINE_BELOW",
                      "LPIPM_MERGE_EDGE", "LPIPM_OVERLAP", "LPIPM_OVERLAP_CUS", "LPIPM_LOOKAHEAD", "LPIPM_LOOKAHEAD_CUS",
                      "LPIPM_GRAPH", "LPIPM_SPECULATE", "LPIPM_HALVES", "LPIPM_ADAT_UNITS", "LPIPM_VEC_FUSED", "LPIPM_STATUS_COPY")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    leaked = [k for k in BIT_CHANGING_KNOBS if k in os.environ]
    if leaked:
        raise pytest.UsageError(f"parity is claimed for the library's defaults: unset {leaked} before running the tests")


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

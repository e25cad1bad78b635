import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# test hooks of the library and its hosts (chunk sizes, pool sizes, forced paths: csrc/sdt_knobs.h) are honoured only under
# SDT_TEST_HOOKS=1; every child process of the suite inherits it
os.environ["SDT_TEST_HOOKS"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    p = ge.load_package()
    if not os.path.exists(p.LIB_PATH):      # hipcc cross-compiles gfx950 without a GPU
        p.build()
    return p


@pytest.fixture(scope="session")
def synth(pkg):
    from soapdenovo_trans_amd import synth as s
    return s

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The built libraries are git-ignored; a fresh checkout builds them once
    (hipcc cross-compiles gfx950 without a GPU; ~15 s)."""
    import nka_amd
    if not os.path.exists(nka_amd.lib_path()):
        nka_amd.build()


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (checker).  Built on demand with gcc."""
    from oracle import oracle_py
    oracle_py.lib()
    return oracle_py

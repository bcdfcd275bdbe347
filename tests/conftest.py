import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (dsvt-ai-trt_amd/ loaded as dsvt_ai_trt_amd).  Loading it dlopens
    libdsvt_hip.so; it is built here if the snapshot lacks it."""
    import __graft_entry__ as g
    so = os.path.join(g.PKG_DIR, "libdsvt_hip.so")
    if not os.path.exists(so):
        g.build()
    return g.load_package()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O

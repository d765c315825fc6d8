import os
import sys

import pytest

# PyTorch bundles its own libamdhip64; libfrr_hip.so links the system one (same SONAME).  Whichever
# loads first serves both, and torch's lazy device init fails if the system runtime got there first,
# so processes that use both import torch first (bench.py does the same).
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): builds oracle/libfrr_oracle.so on first use."""
    from oracle import cref
    cref.lib()
    return cref

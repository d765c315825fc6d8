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


def assert_depth_equal(got, want, err_msg="depth bits differ"):
    """Depth buffers must agree bit for bit, except that a NaN only has to be a NaN: which quiet-NaN pattern 0 * inf
    produces is the property of the machine that ran it (x86 gives 0xFFC00000, gfx950 0x7FC00000), not of the algorithm."""
    import numpy as np
    got = np.asarray(got, np.float32).ravel()
    want = np.asarray(want, np.float32).ravel()
    gn, wn = np.isnan(got), np.isnan(want)
    np.testing.assert_array_equal(gn, wn, err_msg=err_msg + " (NaN pixels)")
    np.testing.assert_array_equal(got.view(np.uint32)[~gn], want.view(np.uint32)[~wn], err_msg=err_msg)


def owned_pixel_rows(H, rank, world, blocked):
    """bool[H]: the pixel rows rank `rank` of `world` owns (both layouts of frr_set_partition_layout)."""
    import numpy as np
    from f_renderer_amd.multigpu import tile_row_owner
    owner = np.asarray(tile_row_owner((H + 31) // 32, world, blocked))
    return owner[np.arange(H) // 32] == rank

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kv-match_amd"), os.path.join(ROOT, "oracle"), os.path.dirname(__file__)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build the oracle (always) and the HIP library (cross-compiles without a GPU) once per session."""
    import __graft_entry__ as ge
    ge.build()
    return True


@pytest.fixture(scope="session")
def series_small(built):
    from kvmatch import datagen
    return datagen.gen_series(200_000, seed=20240501)


def rel_close(a, b, rel=1e-6, abs_floor=1e-9):
    """distances within 1e-6 relative, with an absolute floor for ~0."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.all(np.abs(a - b) <= np.maximum(rel * np.maximum(np.abs(a), np.abs(b)), abs_floor))

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
    config.addinivalue_line("markers", "strict_sort: run with the reference's behaviour where NormQueryEngineDtw's sort throws")


@pytest.fixture(scope="session")
def built():
    """Build the oracle (always) and the HIP library (cross-compiles without a GPU) once per session."""
    import __graft_entry__ as ge
    ge.build()
    return True


@pytest.fixture(autouse=True)
def lenient_sort(request):
    """NormQueryEngineDtw's Arrays.sort throws on a good part of the noisy synthetic queries the GPU tests use (DESIGN.md
    section 2); the reference's query() dies there, and by default so do the oracle and the library.  What the GPU parity
    tests are after is the verification arithmetic, so they run with the switch that carries on in index order -- on BOTH
    sides (oracle.set_lenient_sort / OPT_SORT_VIOLATION on every context a test opens).  Tests marked `strict_sort` keep
    the reference's behaviour: they are the ones that check the throw itself."""
    if "gpu" not in request.keywords or request.node.get_closest_marker("strict_sort"):
        yield
        return
    import kvmatch
    import oracle
    # contexts of module-scoped fixtures exist before this function-scoped one runs
    shared = [request.getfixturevalue(name)[3] for name in ("world", "big") if name in request.fixturenames]
    for c in shared:
        c.set_option(kvmatch.OPT_SORT_VIOLATION, 1)
    orig = kvmatch.Phase2Context.__init__

    def opened_lenient(self, *a, **k):
        orig(self, *a, **k)
        self.set_option(kvmatch.OPT_SORT_VIOLATION, 1)

    kvmatch.Phase2Context.__init__ = opened_lenient
    prev = oracle.set_lenient_sort(True)
    try:
        yield
    finally:
        kvmatch.Phase2Context.__init__ = orig
        oracle.set_lenient_sort(prev)
        for c in shared:
            c.set_option(kvmatch.OPT_SORT_VIOLATION, 0)


@pytest.fixture(scope="session")
def series_small(built):
    from kvmatch import datagen
    return datagen.gen_series(200_000, seed=20240501)


def rel_close(a, b, rel=1e-6, abs_floor=1e-9):
    """distances within 1e-6 relative, with an absolute floor for ~0."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.all(np.abs(a - b) <= np.maximum(rel * np.maximum(np.abs(a), np.abs(b)), abs_floor))

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
    config.addinivalue_line("markers", "lenient_sort: answer cNSM-DTW queries on which the reference's Arrays.sort throws (index order, both sides)")
    config.addinivalue_line("markers", "strict_sort: (kept for readability) the test is about the reference's default, which every unmarked test runs")


@pytest.fixture(scope="session")
def built():
    """Build the oracle (always) and the HIP library (cross-compiles without a GPU) once per session."""
    import __graft_entry__ as ge
    ge.build()
    return True


@pytest.fixture(autouse=True)
def lenient_sort(request):
    """The GPU suite runs the reference's DEFAULT behaviour: where NormQueryEngineDtw's Arrays.sort throws (its comparator
    breaks the contract) the query dies -- in the oracle and in the library.  Queries of cNSM-DTW tests are
    therefore chosen so that JDK 8's TimSort comes through (`sortable_query` below).  Only tests marked `lenient_sort` -- the
    ones that are ABOUT such queries, or that draw random queries by the hundred -- flip the switch that carries on in
    index order, on BOTH sides (oracle.set_lenient_sort / OPT_SORT_VIOLATION on every context the test opens)."""
    if "gpu" not in request.keywords or not request.node.get_closest_marker("lenient_sort"):
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


def sortable_query(series, m, start, step=997, tries=400):
    """First offset >= `start` (walking in steps) whose length-m slice the reference CAN run as a cNSM-DTW query: JDK 8's
    TimSort under the truncating comparator of K/NormQueryEngineDtw.java:475-478 does not throw on it
    (kvm_cnsm_dtw_order returns OK).  Returns the 1-based offset."""
    import kvmatch
    qo = int(start)
    for _ in range(tries):
        if qo + m - 1 > len(series):
            qo = max(1, qo % (len(series) - m))
        try:
            kvmatch.cnsm_dtw_order(series[qo - 1: qo - 1 + m])
            return qo
        except kvmatch.KvMatchIllegalArgument:
            qo += step
    raise AssertionError("no sortable query found")


@pytest.fixture(scope="session")
def series_small(built):
    from kvmatch import datagen
    return datagen.gen_series(200_000, seed=20240501)


def rel_close(a, b, rel=1e-6, abs_floor=1e-9):
    """distances within 1e-6 relative, with an absolute floor for ~0."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.all(np.abs(a - b) <= np.maximum(rel * np.maximum(np.abs(a), np.abs(b)), abs_floor))

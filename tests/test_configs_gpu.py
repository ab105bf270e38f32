"""-m gpu: the two BASELINE.json configurations the other full-size tests do not reach.

configs[0]  "RSM-ED QueryEngine on local-file data-1000000, m=1024, eps=10": the series goes through the reference's own
            file format (files/data-{N}: flat big-endian fp64, K/DataGenerator.java:102-112,
            K/operator/file/TimeSeriesFileOperator.java:54-96) -- written, loaded from the file by the library, read back
            through readTimeSeries by both sides -- and EVERY offset is a candidate, so the engine's semantics equal the
            full scan's (K/CsvTester.java).
configs[4]  "n=10^9 series, mixed ED+DTW query stream": n = 1e9 points (8 GB series + 16 GB prefix sums in HBM), all
            four variants on phase-1-shaped candidate lists spread over the whole series, then a mixed stream that
            alternates them on one context; bit for bit against the oracle.  Needs ~25 GB of host memory (skipped with the
            reason otherwise)."""
import os

import numpy as np
import pytest

import oracle
from conftest import sortable_query

pytestmark = pytest.mark.gpu


def same(got, want):
    return bool(len(got) == len(want[0]) and np.array_equal(got.offsets, want[0]) and np.array_equal(got.distances, want[1])
                and got.cnt_candidate == want[2]["cnt_candidate"])


def test_config_0_file_data_1000000_every_offset(built, tmp_path):
    import kvmatch
    from kvmatch import datagen
    n, m, eps = 1_000_000, 1024, 10.0
    s = datagen.gen_series(n, seed=20241003)
    qoff = sortable_query(s, m, 123_456)     # (a query the reference can run through all four engines)
    planted = [400_000, 777_777, n - m + 1]
    q = datagen.plant_copies(s, qoff, m, planted, [1.0, 1.0, 1.0], [0.0, 0.05, -0.02], 0.01)
    path = tmp_path / f"data-{n}"
    datagen.write_series_file(path, s)
    assert os.path.getsize(path) == n * 8
    raw = np.fromfile(path, dtype=">f8", count=5)
    assert np.array_equal(raw.astype(np.float64), s[:5])                      # big-endian fp64, offset p at byte (p-1)*8
    L, R = np.array([1], dtype=np.int32), np.array([n - m + 1], dtype=np.int32)
    with kvmatch.Phase2Context(0) as ctx:
        ctx.load_series_file(path, n)
        # readTimeSeries(left, length) from the staged file == the reference's file operator on the same file
        for left, length in ((1, 1000), (qoff, m), (n - 999, 1000), (999, 1003)):
            assert np.array_equal(ctx.read_time_series(left, length), oracle.read_time_series_file(str(path), n, left, length))
        want = oracle.rsm_ed(s, q, L, R, 0, eps)
        eng = kvmatch.QueryEngine(ctx)
        for ab in (1, 0):
            ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
            got = eng.verify((L, R), 1, q, eps)
            assert same(got, want), ab
            assert got.stats[0] == n - m + 1
        ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 1)
        offs = set(want[0].tolist())
        assert want[0][0] == qoff and want[1][0] == 0.0                        # the self match (README.md:73)
        assert all(p in offs for p in planted)
        # the same query through the reference's interval path: phase-1-shaped list, lastSegment = 2 (shift 25)
        L2, R2 = datagen.gen_intervals(n, m, 200_000, 64.0, must_cover=[qoff] + planted, seed=9)
        L2s, R2s = (L2 + 25).astype(np.int32), (R2 + 25).astype(np.int32)
        assert same(eng.verify((L2s, R2s), 2, q, eps), oracle.rsm_ed(s, q, L2s, R2s, 25, eps))
        # and the other three engines on the file-staged series (every offset for the ED variant, the list for DTW)
        assert same(kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, 6.0, 1.5, 5.0), oracle.cnsm_ed(s, q, L, R, 0, 6.0, 1.5, 5.0))
        assert same(kvmatch.QueryEngineDtw(ctx).verify((L2, R2), 1, q, eps, 0.05), oracle.rsm_dtw(s, q, L2, R2, 0, eps, 51))
        assert same(kvmatch.NormQueryEngineDtw(ctx).verify((L2, R2), 1, q, 6.0, 0.05, 1.5, 5.0),
                    oracle.cnsm_dtw(s, q, L2, R2, 0, 6.0, 51, 1.5, 5.0))


def test_config_4_n_1e9_all_variants_and_mixed_stream(built):
    import psutil
    import kvmatch
    from kvmatch import datagen
    n = 1_000_000_000
    avail = psutil.virtual_memory().available
    if avail < n * 8 * 2.5:
        pytest.skip(f"host has {avail / 2**30:.0f} GiB available; the n = 1e9 series needs ~20 GiB on the host side")
    s = datagen.gen_series(n)
    qoff = n - 123_456_789
    with kvmatch.Phase2Context(0) as ctx:
        cases = {}
        for m, cand in ((8192, 1_000_000), (4096, 100_000)):
            planted = [1, n // 2, n - m + 1 - 10]
            q = datagen.plant_copies(s, qoff, m, planted, [1.0, 1.0, 1.0], [0.0, 0.02, -0.03], 0.01)
            ctx.load_series(s)
            L, R = datagen.gen_intervals(n, m, cand, 64.0, must_cover=[qoff] + planted, seed=1)
            assert R[-1] > n - 2 * m                                            # the list reaches the far end (int32 offsets ~1e9)
            if m == 8192:
                runs = [("rsm_ed", lambda: kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 10.0), oracle.rsm_ed(s, q, L, R, 0, 10.0)),
                        ("cnsm_ed", lambda: kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, 3.0, 2.0, 10.0),
                         oracle.cnsm_ed(s, q, L, R, 0, 3.0, 2.0, 10.0))]
            else:
                runs = [("rsm_dtw", lambda: kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, 10.0, 0.05),
                         oracle.rsm_dtw(s, q, L, R, 0, 10.0, 204)),
                        ("cnsm_dtw", lambda: kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, 3.0, 0.05, 2.0, 10.0),
                         oracle.cnsm_dtw(s, q, L, R, 0, 3.0, 204, 2.0, 10.0))]
            for name, fn, want in runs:
                got = fn()
                assert same(got, want), name
                assert qoff in set(want[0].tolist()), name
                cases[name] = (fn, want)
            if m == 4096:
                # mixed ED + DTW stream on the staged series (the m = 4096 plant is the one in HBM now): both DTW engines
                # and both ED engines with the shorter query, interleaved, three rounds, identical answers every time
                qe = q
                ed = [("rsm_ed", lambda: kvmatch.QueryEngine(ctx).verify((L, R), 1, qe, 10.0), oracle.rsm_ed(s, qe, L, R, 0, 10.0)),
                      ("cnsm_ed", lambda: kvmatch.NormQueryEngine(ctx).verify((L, R), 1, qe, 3.0, 2.0, 10.0),
                       oracle.cnsm_ed(s, qe, L, R, 0, 3.0, 2.0, 10.0))]
                stream = [ed[0], runs[0], ed[1], runs[1]]
                for _ in range(3):
                    for name, fn, want in stream:
                        assert same(fn(), want), "stream " + name
        assert np.array_equal(ctx.read_time_series(n - 999, 1000), s[n - 1000:])

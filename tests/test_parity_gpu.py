"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same seeded
inputs.  Bar (north_star): identical answer offsets and #answers, distances within 1e-6 relative.  What is asserted
is stronger: cntCandidate identical, answer list identical in content AND order, distances BIT-identical for all four
variants (kvm_exact.hip replays the reference's running sums and sequential additions for every survivor)."""
import os

import numpy as np
import pytest

import oracle
from conftest import sortable_query,  rel_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world(built):
    import kvmatch
    from kvmatch import datagen
    n = 300_000
    series = datagen.gen_series(n, seed=777)
    ctx = kvmatch.Phase2Context(0)
    ctx.load_series(series)
    yield kvmatch, datagen, series, ctx
    ctx.close()


def compare(got, want, exact=True, tag="", abs_floor=1e-9):
    """exact=True (default): offsets, their order and the fp64 distances are bit-identical to the oracle's -- the
    final pass of the library re-evaluates every survivor in the reference's own operation order."""
    o_off, o_dist, o_st = want
    assert got.cnt_candidate == o_st["cnt_candidate"], tag
    assert len(got) == len(o_off), (tag, len(got), len(o_off))
    assert sorted(got.offsets.tolist()) == sorted(o_off.tolist()), tag
    od = dict(zip(o_off.tolist(), o_dist.tolist()))
    ref = np.array([od[o] for o in got.offsets.tolist()])
    if exact:
        assert np.array_equal(ref, got.distances), tag
        assert np.array_equal(got.offsets, o_off), tag  # identical order too
    else:
        assert rel_close(got.distances, ref, abs_floor=abs_floor), tag
    assert np.all(np.diff(got.distances) >= 0), tag  # sorted by distance (K/QueryEngine.java:373)
    assert got.stats[0] == o_st["windows"], tag


def planted(world, m, qoff=123_456, seed=11):
    """Series with the query (the slice at `planted.qoff`: the first offset from `qoff` on that the reference can run as a
    cNSM-DTW query too -- its Arrays.sort does not throw on it, conftest.sortable_query) and three noisy copies."""
    kvmatch, datagen, series, ctx = world
    s = series.copy()
    qoff = planted.qoff = sortable_query(s, m, qoff)
    q = datagen.plant_copies(s, qoff, m, [40_000, 200_000, 250_000], [1.0, 1.2, 0.8], [0.0, 0.5, -1.0], 0.02,
                             seed=seed)
    ctx.load_series(s)
    L, R = datagen.gen_intervals(len(s), m, 30_000, 48.0, must_cover=[qoff, 40_000, 200_000, 250_000], seed=seed)
    return s, q, L, R


@pytest.mark.parametrize("m,eps", [(64, 2.0), (1000, 6.0), (1024, 10.0), (4097, 12.0)])
def test_rsm_ed(world, m, eps):
    kvmatch = world[0]
    s, q, L, R = planted(world, m)
    got = kvmatch.QueryEngine(world[3]).verify((L, R), 1, q, eps)
    compare(got, oracle.rsm_ed(s, q, L, R, 0, eps), tag=f"rsm_ed m={m}")
    assert got.offsets[0] == planted.qoff and got.distances[0] == 0.0  # self-match, README.md:73
    assert len(got) >= 2


@pytest.mark.parametrize("m,eps,alpha,beta", [(128, 3.0, 1.5, 2.0), (1024, 8.0, 1.1, 1.0), (2500, 10.0, 2.0, 10.0)])
def test_cnsm_ed(world, m, eps, alpha, beta):
    kvmatch = world[0]
    s, q, L, R = planted(world, m)
    got = kvmatch.NormQueryEngine(world[3]).verify((L, R), 1, q, eps, alpha, beta)
    want = oracle.cnsm_ed(s, q, L, R, 0, eps, alpha, beta)
    compare(got, want, tag=f"cnsm_ed m={m}")
    assert got.stats[1] == want[2]["pass_filter"]
    assert planted.qoff in got.offsets.tolist() and len(got) >= 2


# rho picks the DP kernel form: one candidate per wave with V = 1 (rho <= 63), 2, 4, 8 (<= 511) cells per lane, LDS form above;
# packed form (dp_form 3, or auto on long lists): 32 lanes x V = 3 .. 8 for rho = 64 .. 255, 64 lanes x V = 5 .. 7 for 256 .. 447
@pytest.mark.parametrize("m,eps,rho", [(64, 2.0, 3), (256, 4.0, 0.05), (512, 6.0, 25), (300, 5.0, 0), (129, 3.0, 200),
                                       (257, 4.0, 127), (1000, 8.0, 100), (1031, 8.0, 77), (1500, 10.0, 300),
                                       (1200, 9.0, 511), (1200, 9.0, 600), (100, 3.0, 63), (100, 3.0, 64),
                                       (2048, 10.0, 0.05), (2300, 10.0, 127), (2300, 10.0, 128), (3000, 12.0, 0.05), (2300, 10.0, 191), (2300, 10.0, 192), (2300, 10.0, 255), (2300, 10.0, 256), (2600, 10.0, 447), (2600, 10.0, 448)])
@pytest.mark.parametrize("dp_form", [0, 3])   # 3: the packed register form for every list length (auto keeps short lists off it)
def test_rsm_dtw(world, m, eps, rho, dp_form):
    kvmatch = world[0]
    world[3].set_option(kvmatch.OPT_DP_KERNEL, dp_form)
    try:
        _rsm_dtw(world, m, eps, rho)
    finally:
        world[3].set_option(kvmatch.OPT_DP_KERNEL, 0)


def _rsm_dtw(world, m, eps, rho):
    kvmatch = world[0]
    s, q, L, R = planted(world, m)
    loose = oracle.resolve_rho(rho, m) >= m
    if loose:
        # the reference dies in lowerUpperLemire over the query (`u[i - r - 1]`, K/utils/DtwUtils.java:81-90): so do both sides by
        # default; the band's natural extension is behind OPT_STRICT_BOUNDS = 0 / oracle.set_strict_bounds(False)
        with pytest.raises(kvmatch.KvMatchIndexOutOfBounds, match=str(m - oracle.resolve_rho(rho, m) - 1)):
            kvmatch.QueryEngineDtw(world[3]).verify((L, R), 1, q, eps, rho)
        with pytest.raises(oracle.OracleError) as ei:
            oracle.rsm_dtw(s, q, L, R, 0, eps, oracle.resolve_rho(rho, m))
        assert ei.value.code == -8
        world[3].set_option(kvmatch.OPT_STRICT_BOUNDS, 0)
        oracle.set_strict_bounds(False)
    try:
        got = kvmatch.QueryEngineDtw(world[3]).verify((L, R), 1, q, eps, rho)
        want = oracle.rsm_dtw(s, q, L, R, 0, eps, oracle.resolve_rho(rho, m))
    finally:
        world[3].set_option(kvmatch.OPT_STRICT_BOUNDS, 1)
        oracle.set_strict_bounds(True)
    compare(got, want, exact=True, tag=f"rsm_dtw m={m} rho={rho}")
    assert planted.qoff in got.offsets.tolist()


def test_nan_alpha_is_answered_like_the_engines_answer_it(world):
    """The REPLs refuse `alpha <= 0.0` (K/NormQueryEngine.java:139-142) -- a NaN passes that test, and in the engine neither
    `std / stdQ <= alpha` nor `>= 1 / alpha` holds for it: no answers, cntCandidate as ever (rounds 1-2 refused it)."""
    kvmatch, datagen, series, ctx = world
    m = 128
    qoff = sortable_query(series, m, 40_000)
    q = series[qoff - 1: qoff - 1 + m].copy()
    L, R = datagen.gen_intervals(len(series), m, 5_000, 20.0, must_cover=[qoff], seed=9)
    nan = float("nan")
    compare(kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, 3.0, nan, 2.0), oracle.cnsm_ed(series, q, L, R, 0, 3.0, nan, 2.0), tag="nan alpha ed")
    compare(kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, 3.0, 6, nan, 2.0), oracle.cnsm_dtw(series, q, L, R, 0, 3.0, 6, nan, 2.0), tag="nan alpha dtw")
    with pytest.raises(kvmatch.KvMatchIllegalArgument):
        kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, 3.0, 0.0, 2.0)


def test_dtw_engines_die_where_the_reference_does(world):
    """DtwUtils.lowerUpperLemire stores `u[i - r - 1]` from i = len on (K/utils/DtwUtils.java:81-90): over an array of
    len <= r points that is ArrayIndexOutOfBoundsException, uncaught, and query() dies -- for the query envelope when
    rho >= m (K/QueryEngineDtw.java:362, K/NormQueryEngineDtw.java:469: before the sort, before anything is read), for the
    data envelope when a fetched range holds <= rho points (:399 / :524: a stub beyond the last window offset), right behind
    the readTimeSeries that throws IllegalArgumentException for an EMPTY range -- whichever interval offends first decides.
    KVM_OPT_STRICT_BOUNDS (default 1) mirrors that; 0 computes the envelope's natural extension, as rounds 1-2 did."""
    kvmatch, datagen, series, ctx = world
    n, m, rho = len(series), 100, 30
    s = series
    ctx.load_series(s)
    q = s[5_000:5_000 + m].copy()
    eds = (("rsm", lambda iv, r: kvmatch.QueryEngineDtw(ctx).verify(iv, 1, q, 3.0, r, resolved=True),
            lambda iv, r: oracle.rsm_dtw(s, q, iv[0], iv[1], 0, 3.0, r)),
           ("cnsm", lambda iv, r: kvmatch.NormQueryEngineDtw(ctx).verify(iv, 1, q, 3.0, r, 1.5, 5.0, resolved=True),
            lambda iv, r: oracle.cnsm_dtw(s, q, iv[0], iv[1], 0, 3.0, r, 1.5, 5.0)))
    arr = lambda *v: np.array(v, dtype=np.int32)
    good = (arr(4_990), arr(5_010))
    stub_oob = (arr(4_990, n - 20), arr(5_010, n - 15))     # second interval fetches [n-20, n]: 21 points <= rho
    stub_ok = (arr(4_990, n - 60), arr(5_010, n - 55))      # 61 points: longer than the band, shorter than m -> no window, no error
    empty_first = (arr(4_990, n + 5, n - 20), arr(5_010, n + 9, n - 15))   # an empty range comes first: IllegalArgument wins
    oob_first = (arr(4_990, n - 20, n + 5), arr(5_010, n - 15, n + 9))
    for name, lib, orc in eds:
        # rho >= m
        for r in (m, m + 7, 3 * m):
            with pytest.raises(kvmatch.KvMatchIndexOutOfBounds, match=f"^{m - r - 1}$"):
                lib(good, r)
            with pytest.raises(oracle.OracleError) as ei:
                orc(good, r)
            assert ei.value.code == -8, name
        compare(lib(good, m - 1), orc(good, m - 1), tag=f"{name} rho = m - 1")
        # ... and that comes before the list is looked at: an unsorted list (refused as such by the cNSM entry points, sorted
        # by the RSM ones) does not change the exception (round 3: the oracle's guard stood in front of the envelope)
        unsorted = (arr(5_100, 4_990), arr(5_120, 5_010))
        with pytest.raises(kvmatch.KvMatchIndexOutOfBounds):
            lib(unsorted, m + 7)
        with pytest.raises(oracle.OracleError) as ei:
            orc(unsorted, m + 7)
        assert ei.value.code == -8, name
        # stubs
        with pytest.raises(kvmatch.KvMatchIndexOutOfBounds, match="^-10$"):
            lib(stub_oob, rho)
        with pytest.raises(oracle.OracleError) as ei:
            orc(stub_oob, rho)
        assert ei.value.code == -8, name
        compare(lib(stub_ok, rho), orc(stub_ok, rho), tag=f"{name} stub longer than the band")
        if name == "rsm":   # (the cNSM engines merge the three intervals into one chunk: no empty read there)
            with pytest.raises(kvmatch.KvMatchIllegalArgument):
                lib(empty_first, rho)
            with pytest.raises(oracle.OracleError) as ei:
                orc(empty_first, rho)
            assert ei.value.code == -2
            with pytest.raises(kvmatch.KvMatchIndexOutOfBounds):
                lib(oob_first, rho)
            with pytest.raises(oracle.OracleError) as ei:
                orc(oob_first, rho)
            assert ei.value.code == -8
        # the natural extension behind the switch: same answers on both sides
        ctx.set_option(kvmatch.OPT_STRICT_BOUNDS, 0)
        oracle.set_strict_bounds(False)
        try:
            compare(lib(good, 3 * m), orc(good, 3 * m), tag=f"{name} rho = 3 m, not strict")
            compare(lib(stub_oob, rho), orc(stub_oob, rho), tag=f"{name} stub, not strict")
        finally:
            ctx.set_option(kvmatch.OPT_STRICT_BOUNDS, 1)
            oracle.set_strict_bounds(True)


@pytest.mark.parametrize("m,eps,rho,alpha,beta", [(128, 3.0, 6, 1.5, 2.0), (512, 6.0, 0.05, 1.3, 1.5),
                                                  (1024, 8.0, 51, 2.0, 10.0), (1500, 9.0, 100, 1.5, 5.0),
                                                  (3000, 12.0, 0.05, 2.0, 10.0), (2600, 11.0, 300, 1.5, 5.0)])
@pytest.mark.parametrize("dp_form", [0, 3])
def test_cnsm_dtw(world, m, eps, rho, alpha, beta, dp_form):
    kvmatch = world[0]
    s, q, L, R = planted(world, m)
    world[3].set_option(kvmatch.OPT_DP_KERNEL, dp_form)
    try:
        got = kvmatch.NormQueryEngineDtw(world[3]).verify((L, R), 1, q, eps, rho, alpha, beta)
    finally:
        world[3].set_option(kvmatch.OPT_DP_KERNEL, 0)
    want = oracle.cnsm_dtw(s, q, L, R, 0, eps, oracle.resolve_rho(rho, m), alpha, beta)
    compare(got, want, tag=f"cnsm_dtw m={m}")
    assert got.stats[1] == want[2]["pass_filter"]
    assert planted.qoff in got.offsets.tolist() and len(got) >= 2


@pytest.mark.parametrize("rho", [0, 1, 7, 64, 204, 1500, 5000])
def test_data_envelope_equals_lower_upper_lemire(world, rho):
    """kvm_envelope_kernel (sparse-table sliding max / min in LDS) against the reference's deque algorithm restated in the
    oracle (DtwUtils.lowerUpperLemire + CircularArray): every point, including the clipped ends, on data with long constant
    stretches and repeated values (ties in the deque's comparisons)."""
    kvmatch, datagen, series, ctx = world
    s = series[:120_000].copy()
    s[5_000:9_000] = 2.5                                  # a constant stretch longer than most radii
    s[20_000:20_700] = np.round(s[20_000:20_700])         # many equal neighbours
    ctx.load_series(s)
    u, l = ctx.data_envelope(rho)
    wl, wu = oracle.lower_upper_lemire(s, rho)
    assert np.array_equal(u, wu) and np.array_equal(l, wl)
    u2, l2 = ctx.data_envelope(rho, first=777, count=1234)
    assert np.array_equal(u2, wu[776:776 + 1234]) and np.array_equal(l2, wl[776:776 + 1234])


def test_query_stream_three_in_flight_matches_sequential(world):
    """kvm_share_series + QueryStream: three contexts over one resident series, driven by three host threads, return for a
    mixed stream of all four variants exactly what one context returns sequentially (and what the oracle returns)."""
    kvmatch = world[0]
    ctx = world[3]
    s, q, L, R = planted(world, 300)
    jobs_spec = [("rsm_ed", (5.0,)), ("cnsm_dtw", (4.0, 15, 1.5, 2.0)), ("cnsm_ed", (4.0, 1.5, 2.0)), ("rsm_dtw", (5.0, 15))] * 6
    cls = {"rsm_ed": kvmatch.QueryEngine, "cnsm_ed": kvmatch.NormQueryEngine, "rsm_dtw": kvmatch.QueryEngineDtw,
           "cnsm_dtw": kvmatch.NormQueryEngineDtw}
    want = {"rsm_ed": oracle.rsm_ed(s, q, L, R, 0, 5.0), "cnsm_ed": oracle.cnsm_ed(s, q, L, R, 0, 4.0, 1.5, 2.0),
            "rsm_dtw": oracle.rsm_dtw(s, q, L, R, 0, 5.0, 15), "cnsm_dtw": oracle.cnsm_dtw(s, q, L, R, 0, 4.0, 15, 1.5, 2.0)}
    seen = set()

    def job(name, args):
        def run(c):
            seen.add(id(c))
            return cls[name](c).verify((L, R), 1, q, *args)
        return run

    with kvmatch.QueryStream(ctx, in_flight=3) as qs:
        assert len(qs.ctxs) == 3 and all(c.n == ctx.n for c in qs.ctxs)
        res = qs.run([job(n, a) for n, a in jobs_spec])
        for (name, _), got in zip(jobs_spec, res):
            compare(got, want[name], tag="stream " + name)
        assert len(seen) >= 2                      # the jobs really were spread over several contexts
        with pytest.raises(kvmatch.KvMatchIllegalArgument):
            ctx.share_series(ctx)                  # a context cannot share its own series
    # the owner is untouched after the sharing contexts are closed
    compare(kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 5.0), want["rsm_ed"], tag="owner after close")


@pytest.mark.parametrize("scale,jump", [(1e-158, 1.0), (1.0, 1e12), (1e-3, 1e6)])
def test_windows_ending_just_in_front_of_a_burst(world, scale, jump):
    """The prefix sums of the 1-3 points in front of a value many orders of magnitude above the local scale (they share a
    4-point group with it in kvm_prefix_kernel) must not feel it: round 2 found `inclusive - own total` there, which wiped
    out their prefix and made the (alpha, beta) filter drop answers whose windows END right in front of the burst.  Exact
    copies of the query are planted so that they end 1, 2 and 3 points in front of three bursts."""
    kvmatch, datagen, series, ctx = world
    rng = np.random.default_rng(5)
    n, m = 40_000, 512
    s = np.cumsum(rng.uniform(-1, 1, n)) * scale
    q = s[9_000:9_000 + m].copy()
    copies = []
    for k, g in ((1, 20_003), (2, 24_007), (3, 28_011)):        # g = 0-based index of the burst's first point, g % 4 == 3
        assert g % 4 == 3
        s[g: g + 700] += jump
        off = g - k - m + 2                                     # 1-based start of a window whose last point is 0-based g - k
        s[off - 1: off - 1 + m] = q
        copies.append(off)
    ctx.load_series(s)
    eps = 0.5 * np.sqrt(m)
    # (a) phase-1-shaped: one short interval per copy, so that the reference's running sums start fresh in front of it
    #     (in one long run they would still carry the rounding residue of the previous burst)
    L = np.array([c - 40 for c in copies] + [8_990], dtype=np.int32)
    R = np.array([c + 1 for c in copies] + [9_010], dtype=np.int32)
    order = np.argsort(L)
    L, R = L[order], R[order]
    got = kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, eps, 1.5, 1e9)
    compare(got, oracle.cnsm_ed(s, q, L, R, 0, eps, 1.5, 1e9), tag=f"burst scale={scale} jump={jump}")
    assert all(c in set(got.offsets.tolist()) for c in copies) and 9_001 in set(got.offsets.tolist())
    got = kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, eps, 25, 1.5, 1e9)
    compare(got, oracle.cnsm_dtw(s, q, L, R, 0, eps, 25, 1.5, 1e9), tag=f"burst dtw scale={scale}")
    assert all(c in set(got.offsets.tolist()) for c in copies)
    # (b) every offset in one run: whatever the reference's drifting sums make of it, the library makes the same
    L, R = np.array([1], dtype=np.int32), np.array([n - m + 1], dtype=np.int32)
    for alpha, beta in ((1.5, 1e9), (1.0000001, 1e-6 * scale + 1e-300)):
        compare(kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, eps, alpha, beta), oracle.cnsm_ed(s, q, L, R, 0, eps, alpha, beta),
                tag=f"burst full scale={scale} jump={jump} a={alpha}")


def test_all_offsets_equals_full_scan(world):
    """candidates = every offset: engine semantics == CsvTester / UCR full-scan semantics."""
    kvmatch, datagen, series, ctx = world
    s = series[:50_000].copy()
    m = 200
    q = datagen.plant_copies(s, 7_777, m, [30_000], [1.0], [0.0], 0.05)
    ctx.load_series(s)
    L = np.array([1], dtype=np.int32)
    R = np.array([len(s) - m + 1], dtype=np.int32)
    compare(kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 4.0), oracle.rsm_ed(s, q, L, R, 0, 4.0), tag="full rsm_ed")
    compare(kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, 3.0, 1.5, 2.0),
            oracle.cnsm_ed(s, q, L, R, 0, 3.0, 1.5, 2.0), tag="full cnsm_ed")
    compare(kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, 4.0, 10), oracle.rsm_dtw(s, q, L, R, 0, 4.0, 10),
            exact=True, tag="full rsm_dtw")
    compare(kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, 3.0, 10, 1.5, 2.0),
            oracle.cnsm_dtw(s, q, L, R, 0, 3.0, 10, 1.5, 2.0), tag="full cnsm_dtw")


def test_shift_clipping_empty_and_errors(world):
    kvmatch, datagen, series, ctx = world
    s = series[:20_000].copy()
    ctx.load_series(s)
    m = 100
    qo = sortable_query(s, m, 1, step=7)
    q = s[qo - 1:qo - 1 + m].copy()
    left = np.array([10, 5000, 19_950], dtype=np.int32)
    right = np.array([60, 5100, 20_040], dtype=np.int32)
    for last_segment in (1, 3):
        shift = (last_segment - 1) * 25
        compare(kvmatch.QueryEngine(ctx).verify((left, right), last_segment, q, 1e9),
                oracle.rsm_ed(s, q, left, right, shift, 1e9), tag=f"clip ls={last_segment}")
        compare(kvmatch.NormQueryEngine(ctx).verify((left, right), last_segment, q, 1e9, 100.0, 1e9),
                oracle.cnsm_ed(s, q, left, right, shift, 1e9, 100.0, 1e9), tag=f"clip cnsm ls={last_segment}")
    e = np.empty(0, dtype=np.int32)
    for eng, args in ((kvmatch.QueryEngine, (1.0,)), (kvmatch.NormQueryEngine, (1.0, 1.5, 1.0)),
                      (kvmatch.QueryEngineDtw, (1.0, 3)), (kvmatch.NormQueryEngineDtw, (1.0, 3, 1.5, 1.0))):
        a = eng(ctx).verify((e, e), 1, q, *args)
        assert len(a) == 0 and a.cnt_candidate == 0
    # interval wholly beyond the series: IllegalArgumentException, as readTimeSeries throws
    with pytest.raises(kvmatch.KvMatchIllegalArgument):
        kvmatch.QueryEngine(ctx).verify([(30_000, 30_010)], 1, q, 1.0)
    with pytest.raises(oracle.OracleError):
        oracle.rsm_ed(s, q, [30_000], [30_010], 0, 1.0)
    # readTimeSeries range errors (K/operator/file/TimeSeriesFileOperator.java:55-57)
    assert np.array_equal(ctx.read_time_series(101, 50), s[100:150])
    for left_, len_ in ((0, 5), (19_999, 3), (5, 0)):
        with pytest.raises(kvmatch.KvMatchIllegalArgument):
            ctx.read_time_series(left_, len_)


def test_knobs_do_not_change_answers(world):
    kvmatch = world[0]
    ctx = world[3]
    for removed in (2, 4, 5, -1):   # (2: the direct LDS-tiled pruner, removed in round 3; 4: the lane-granular one, round 4)
        with pytest.raises(kvmatch.KvMatchIllegalArgument):
            ctx.set_option(kvmatch.OPT_ED_KERNEL, removed)
    assert ctx.get_option(kvmatch.OPT_ED_KERNEL) == 0 and ctx.get_option(kvmatch.OPT_STRICT_BOUNDS) == 1 and ctx.get_option(kvmatch.OPT_UCR_LOOPS) == 0
    s, q, L, R = planted(world, 256)
    base_ed = kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 5.0)
    base_dtw = kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, 4.0, 12, 1.5, 2.0)
    try:
        ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0)
        ctx.set_option(kvmatch.OPT_LB_CASCADE, 0)
        a = kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 5.0)
        b = kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, 4.0, 12, 1.5, 2.0)
    finally:
        ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 1)
        ctx.set_option(kvmatch.OPT_LB_CASCADE, 1)
    assert np.array_equal(a.offsets, base_ed.offsets) and np.array_equal(a.distances, base_ed.distances)
    assert np.array_equal(b.offsets, base_dtw.offsets) and np.array_equal(b.distances, base_dtw.distances)
    ctx.set_option(kvmatch.OPT_DP_KERNEL, 1)  # generic LDS form of the DP
    try:
        b2 = kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, 4.0, 12, 1.5, 2.0)
    finally:
        ctx.set_option(kvmatch.OPT_DP_KERNEL, 0)
    assert np.array_equal(b2.offsets, base_dtw.offsets) and np.array_equal(b2.distances, base_dtw.distances)
    for k in (1, 3):
        ctx.set_option(kvmatch.OPT_ED_KERNEL, k)
        try:
            a = kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 5.0)
            c = kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, 4.0, 1.5, 2.0)
        finally:
            ctx.set_option(kvmatch.OPT_ED_KERNEL, 0)
        assert np.array_equal(a.offsets, base_ed.offsets) and np.array_equal(a.distances, base_ed.distances)
        compare(c, oracle.cnsm_ed(s, q, L, R, 0, 4.0, 1.5, 2.0), tag=f"ed kernel {k}")


def test_series_loaders_agree(world, tmp_path):
    """big-endian bytes / files/data-{N} / native array all stage the same series (a18)."""
    kvmatch, datagen, series, ctx = world
    s = series[:12_345].copy()
    p = tmp_path / "data-12345"
    datagen.write_series_file(p, s)
    raw = p.read_bytes()
    assert raw == oracle.encode_be64(s)
    for loader in (lambda: ctx.load_series(s), lambda: ctx.load_series_be64(raw, len(s)),
                   lambda: ctx.load_series_file(p, len(s))):
        loader()
        assert ctx.n == len(s)
        assert np.array_equal(ctx.read_time_series(1, len(s)), s)
    with pytest.raises(kvmatch.KvMatchIOError):
        ctx.load_series_file(tmp_path / "nope", 10)


def test_attach_torch_device_series(world):
    import torch
    kvmatch, datagen, series, ctx = world
    s = series[:40_000].copy()
    t = torch.from_numpy(s).to("cuda:0")
    ctx2 = kvmatch.Phase2Context(0)
    try:
        ctx2.attach_device_series(t.data_ptr(), t.numel(), keepalive=t)
        m = 128
        q = s[999:999 + m].copy()
        L, R = datagen.gen_intervals(len(s), m, 5000, 16.0, must_cover=[1000], seed=2)
        compare(kvmatch.NormQueryEngine(ctx2).verify((L, R), 1, q, 3.0, 1.5, 2.0),
                oracle.cnsm_ed(s, q, L, R, 0, 3.0, 1.5, 2.0), tag="attached")
    finally:
        ctx2.close()


def test_prefix_sums_accuracy_far_from_origin(world):
    """mu/sigma from block-restarted prefix sums stay accurate deep into a long random-walk stretch
    (SURVEY 7 'hard parts'): cNSM answers late in the series still match the oracle."""
    kvmatch, datagen, series, ctx = world
    rng = np.random.default_rng(5)
    s = np.cumsum(rng.uniform(-1, 1, 400_000)) + 3000.0  # |x| ~ 3000: sum x^2 ~ 1e7 per point
    m = 1024
    qoff = 390_000
    q = datagen.plant_copies(s, qoff, m, [16_000, 100_000], [1.0, 1.1], [0.0, 3.0], 0.01)
    ctx.load_series(s)
    L, R = datagen.gen_intervals(len(s), m, 20_000, 32.0, must_cover=[qoff, 16_000, 100_000, 16_384 - 512], seed=9)
    got = kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, 2.0, 1.5, 5.0)
    # Conditioning: |x| ~ 3000 with window variance ~ 100 makes the REFERENCE's own std = sqrt(ex2/m - mean^2)
    # (K/NormQueryEngine.java:198,508) lose ~1e-10 relative (E[x^2]/var ~ 1e5 ulps); that noise sits in stdQ, which
    # this library computes exactly as the reference does.  A z-normalised self-match therefore sits at
    # ~sqrt(m) * 1e-10 ~ 3e-9 instead of 0 unless both sides round identically (the oracle's self-match does: the
    # query's own window starts its interval).  Absolute floor scaled accordingly; relative tolerance unchanged.
    compare(got, oracle.cnsm_ed(s, q, L, R, 0, 2.0, 1.5, 5.0), tag="far prefix")
    assert len(got) >= 2


def test_full_size_properties_n1e8_m8192(built):
    """BASELINE configs[1]/[2] at full size (n=1e8, m=8192, ~1e6 candidates): size-independent properties instead of
    an oracle pass -- self-match at distance exactly 0, every planted copy found, answers sorted, two independent GPU
    implementations (LDS-tiled vs wave-per-candidate) and early-abandon on/off agree, cNSM self-match ~0 and its
    answers a superset of the planted copies, oracle agreement on a bounded sample of the intervals."""
    import kvmatch
    from kvmatch import datagen
    n, m = 100_000_000, 8192
    series = datagen.gen_series(n)
    qoff = n // 3 + 17
    planted = [n // 7, n // 2 + 1000, n - 3 * m]
    q = datagen.plant_copies(series, qoff, m, planted, [1.0, 1.0, 1.0], [0.0, 0.02, -0.03], 0.01)
    L, R = datagen.gen_intervals(n, m, 1_000_000, 64.0, must_cover=[qoff] + planted, seed=1)
    with kvmatch.Phase2Context(0) as ctx:
        ctx.load_series(series)
        eng = kvmatch.QueryEngine(ctx)
        a = eng.verify((L, R), 1, q, 10.0)
        assert a.offsets[0] == qoff and a.distances[0] == 0.0
        assert set(planted) <= set(a.offsets.tolist()) and np.all(np.diff(a.distances) >= 0)
        assert a.cnt_candidate == int((R.astype(np.int64) - L + 1).sum()) == a.stats[0]
        ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0)
        b = eng.verify((L, R), 1, q, 10.0)
        ctx.set_option(kvmatch.OPT_ED_KERNEL, 1)
        c = eng.verify((L, R), 1, q, 10.0)
        ctx.set_option(kvmatch.OPT_ED_KERNEL, 0)
        ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 1)
        assert np.array_equal(a.offsets, b.offsets) and np.array_equal(a.distances, b.distances)
        assert np.array_equal(a.offsets, c.offsets) and np.array_equal(a.distances, c.distances)
        z = kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, 2.0, 1.5, 5.0)
        assert z.offsets[0] == qoff and z.distances[0] < 1e-6 and set(planted) <= set(z.offsets.tolist())
        # oracle on a bounded sample: the intervals around the query and the planted copies plus the first 300
        pick = np.zeros(len(L), dtype=bool)
        pick[:300] = True
        for o in [qoff] + planted:
            pick |= (L <= o) & (R >= o)
        Ls, Rs = L[pick], R[pick]
        compare(eng.verify((Ls, Rs), 1, q, 10.0), oracle.rsm_ed(series, q, Ls, Rs, 0, 10.0), tag="n1e8 sample rsm")
        compare(kvmatch.NormQueryEngine(ctx).verify((Ls, Rs), 1, q, 2.0, 1.5, 5.0),
                oracle.cnsm_ed(series, q, Ls, Rs, 0, 2.0, 1.5, 5.0), tag="n1e8 sample cnsm")


def test_full_size_dtw_m4096_rho5pct(built):
    """BASELINE configs[3]: cNSM-DTW, m=4096, rho=204 on n=1e7 (the DP is independent of n): cascade on/off and both DP
    kernel forms give identical answers; oracle agreement on a bounded sample."""
    import kvmatch
    from kvmatch import datagen
    n, m = 10_000_000, 4096
    series = datagen.gen_series(n, seed=4242)
    qoff = n // 3 + 5
    planted = [n // 5, n // 2 + 77]
    q = datagen.plant_copies(series, qoff, m, planted, [1.0, 1.1], [0.0, 0.5], 0.05)
    L, R = datagen.gen_intervals(n, m, 200_000, 64.0, must_cover=[qoff] + planted, seed=3)
    with kvmatch.Phase2Context(0) as ctx:
        ctx.load_series(series)
        eng = kvmatch.NormQueryEngineDtw(ctx)
        a = eng.verify((L, R), 1, q, 8.0, 0.05, 1.5, 5.0)
        assert a.offsets[0] == qoff and a.distances[0] < 1e-6 and set(planted) <= set(a.offsets.tolist())
        assert a.stats[1] >= a.stats[2] >= a.stats[3] >= a.stats[4] >= len(a)
        ctx.set_option(kvmatch.OPT_LB_CASCADE, 0)
        pick = np.zeros(len(L), dtype=bool)
        pick[:40] = True
        for o in [qoff] + planted:
            pick |= (L <= o) & (R >= o)
        Ls, Rs = L[pick], R[pick]
        s1 = eng.verify((Ls, Rs), 1, q, 8.0, 0.05, 1.5, 5.0)
        ctx.set_option(kvmatch.OPT_DP_KERNEL, 1)
        s2 = eng.verify((Ls, Rs), 1, q, 8.0, 0.05, 1.5, 5.0)
        ctx.set_option(kvmatch.OPT_DP_KERNEL, 0)
        ctx.set_option(kvmatch.OPT_LB_CASCADE, 1)
        s3 = eng.verify((Ls, Rs), 1, q, 8.0, 0.05, 1.5, 5.0)
        assert np.array_equal(s1.offsets, s2.offsets) and np.array_equal(s1.distances, s2.distances)
        assert np.array_equal(s1.offsets, s3.offsets) and np.array_equal(s1.distances, s3.distances)
        compare(s3, oracle.cnsm_dtw(series, q, Ls, Rs, 0, 8.0, 204, 1.5, 5.0), tag="m4096 sample")


def test_device_resident_verifier_matches_host_api(world):
    """kvm_verify_device (survivors stay in HBM, caller's stream) == kvm_verify_* for all four variants."""
    kvmatch = world[0]
    from kvmatch import distributed as D
    ctx = world[3]
    s, q, L, R = planted(world, 300)
    cands = int((R.astype(np.int64) - L + 1).sum())
    host = [kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 5.0),
            kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, 4.0, 1.5, 2.0),
            kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, 5.0, 15),
            kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, 4.0, 15, 1.5, 2.0)]
    args = [dict(epsilon=5.0), dict(epsilon=4.0, alpha=1.5, beta=2.0), dict(epsilon=5.0, rho=15),
            dict(epsilon=4.0, rho=15, alpha=1.5, beta=2.0)]
    for variant in range(4):
        for pack_cap in (4096, 1):  # 1 forces the exact-size fallback path
            v = D.ShardedVerifier(ctx, variant, cands, pack_cap=pack_cap)
            offs, dists = v.verify((L, R), 1, q, **args[variant])
            assert np.array_equal(offs, host[variant].offsets), variant
            assert np.array_equal(dists, host[variant].distances), variant
            assert v.cnt_candidate == host[variant].cnt_candidate


def test_verify_device_alternating_streams_on_integer_data(world):
    """kvm_verify_device returns without synchronising; for the DTW engines the sequential tie path (integer-valued data,
    eps^2 an achievable value: routinely taken) enqueues its kernels and the copy of the answer count AFTER run_variant's own
    record of ev_staged.  The next call on the same context waits on that event only -- so it has to be re-recorded behind
    everything the call enqueued, or a call on ANOTHER stream may reuse the scratch while the tie path still reads it
    (advisor, round 2).  Two streams alternate on one context; every answer list must equal the host path's."""
    import torch
    kvmatch = world[0]
    from kvmatch import distributed as D
    ctx = world[3]
    rng = np.random.default_rng(4711)
    n, m = 120_000, 64
    s = rng.integers(-3, 4, n).astype(np.float64)
    q = s[60_000:60_000 + m].copy()
    for p in (20_000, 90_000):
        s[p - 1:p - 1 + m] = q
    ctx.load_series(s)
    try:
        L, R = np.array([1], dtype=np.int32), np.array([n - m + 1], dtype=np.int32)
        cands = n - m + 1
        dev = torch.device(f"cuda:{ctx.device}")
        jobs = [(2, dict(epsilon=3.0, rho=4)), (3, dict(epsilon=2.0, rho=4, alpha=1.5, beta=1.0)),
                (2, dict(epsilon=0.0, rho=2)), (3, dict(epsilon=3.0, rho=6, alpha=2.0, beta=2.0))]
        want = [kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, 3.0, 4), kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, 2.0, 4, 1.5, 1.0),
                kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, 0.0, 2), kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, 3.0, 6, 2.0, 2.0)]
        assert sum(w.stats[kvmatch.STAT_DTW_SEQUENTIAL] for w in want) > 0     # the tie path is what this is about
        streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        verifiers = {}
        for rep in range(3):
            launched = []
            for i, (variant, kw) in enumerate(jobs):
                st = streams[(i + rep) % 2]
                with torch.cuda.stream(st):
                    v = verifiers.setdefault((i, (i + rep) % 2), D.ShardedVerifier(ctx, variant, cands))
                    v.launch((L, R), 1, q, kw["epsilon"], kw.get("rho", 0), kw.get("alpha", 1.0), kw.get("beta", 0.0), None)
                    launched.append((v, st))             # (no synchronisation between the calls)
            for st in streams:
                st.synchronize()
            for i, (v, st) in enumerate(launched):
                cnt = int(v.count.item())
                offs, dists = D.sort_answers(v.out_off[:cnt].cpu().numpy(), v.out_dist[:cnt].cpu().numpy())
                assert np.array_equal(offs, want[i].offsets) and np.array_equal(dists, want[i].distances), (rep, i)
    finally:
        ctx.load_series(world[2])


def test_verify_device_on_a_foreign_stream_with_a_fresh_rho(world):
    """kvm_verify_device on a non-blocking, non-default stream: the data envelope of a rho the context has not seen yet
    must be built on (or ordered before) THAT stream -- LB_Keogh on the data envelope reads it in the same call.  Then the
    cached envelope is reused from a second foreign stream (ordered by the event recorded behind the build)."""
    import torch
    kvmatch = world[0]
    from kvmatch import distributed as D
    ctx = world[3]
    s, q, L, R = planted(world, 300)
    cands = int((R.astype(np.int64) - L + 1).sum())
    dev = torch.device(f"cuda:{ctx.device}")
    for rho in (37, 91):  # not used by any other test of this session's context
        host = kvmatch.NormQueryEngineDtw  # host API answers AFTER the device call, so the device call sees a cold cache
        streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        got = []
        for st in streams:
            with torch.cuda.stream(st):
                v = D.ShardedVerifier(ctx, 3, cands)
                got.append(v.verify((L, R), 1, q, 4.0, rho=rho, alpha=1.5, beta=2.0))
        want = host(ctx).verify((L, R), 1, q, 4.0, rho, 1.5, 2.0)
        for offs, dists in got:
            assert np.array_equal(offs, want.offsets), rho
            assert np.array_equal(dists, want.distances), rho
        ref = oracle.cnsm_dtw(s, q, L, R, 0, 4.0, rho, 1.5, 2.0)
        assert np.array_equal(want.offsets, ref[0]) and np.array_equal(want.distances, ref[1])


# ---------------------------------------------------------------------------------------------------------------
# Knife edges: thresholds placed exactly ON a candidate's decision quantity (and one ulp to either side).  These force
# the guard-band -> exact-pass route of the library (kvm_exact.hip); random data never gets there by itself.

def _seq_sum(terms):
    """fp64 additions strictly in order (what the reference's `distance += ...` loop does)."""
    return float(np.add.accumulate(np.asarray(terms, dtype=np.float64))[-1])


def _eps_for_d2(d2):
    """smallest fp64 eps with eps*eps >= d2, and the largest with eps*eps < d2"""
    e = np.sqrt(d2)
    while e * e >= d2:
        e = np.nextafter(e, 0.0)
    below = e
    while e * e < d2:
        e = np.nextafter(e, np.inf)
    return float(e), float(below)


def test_knife_edge_rsm_ed_epsilon(world):
    kvmatch, datagen, series, ctx = world
    s = series[:60_000].copy()
    m = 777
    qoff = 20_000
    q = datagen.plant_copies(s, qoff, m, [5_000, 41_000], [1.0, 1.0], [0.01, -0.02], 0.02, seed=3)
    ctx.load_series(s)
    L, R = datagen.gen_intervals(len(s), m, 8_000, 24.0, must_cover=[qoff, 5_000, 41_000], seed=8)
    for target in (5_000, 41_000):
        w = s[target - 1: target - 1 + m]
        d2 = _seq_sum((w - q) * (w - q))           # the reference's distance^2 of that candidate, bit for bit
        eps_in, eps_out = _eps_for_d2(d2)
        assert eps_in * eps_in >= d2 > eps_out * eps_out
        a = kvmatch.QueryEngine(ctx).verify((L, R), 1, q, eps_in)
        b = kvmatch.QueryEngine(ctx).verify((L, R), 1, q, eps_out)
        compare(a, oracle.rsm_ed(s, q, L, R, 0, eps_in), tag="knife in")
        compare(b, oracle.rsm_ed(s, q, L, R, 0, eps_out), tag="knife out")
        assert target in a.offsets.tolist() and target not in b.offsets.tolist()
        assert a.distances[a.offsets.tolist().index(target)] == np.sqrt(d2)


def test_rsm_ed_abandoning_windows_that_agree_with_the_query_over_their_head(world):
    """Early abandoning on a periodic series whose every in-phase window agrees with the query over its first 1024 points, with
    the differences behind them: nothing dies where the head checks look (kvm_retile_rsm_kernel, the pruner's first chunk),
    waves are left with few or many live candidates (the pruner's HAND-OVER rule, kvm_ed_dot.hip, on both sides of its
    threshold), some windows die late, some end exactly at epsilon (in and out), some are answers."""
    kvmatch, datagen, series, ctx = world
    rng = np.random.default_rng(41)
    m, period = 2304, 256
    n = 120_000
    base = rng.standard_normal(period)
    s = np.tile(base, n // period + 1)[:n].copy()
    qoff = 10 * period + 1
    q = s[qoff - 1: qoff - 1 + m].copy()
    L = np.arange(1, n - m, period, dtype=np.int64)[:400]      # every candidate start in phase with the query's
    R = L + 63                                                    # (and 63 starts out of phase behind each)
    for i, o in enumerate(L.tolist()):                            # spoil the tails: by nothing, a little, or a lot
        kind = i % 4
        if kind == 1:
            s[o - 1 + 1500] += 0.75
        elif kind == 2:
            s[o - 1 + 2303] += 3.0
        elif kind == 3:
            s[o - 1 + 1024] += 40.0
    # (the spoiled points of one candidate sit inside its neighbours' windows too: whatever that makes, the oracle sees the same)
    q = s[qoff - 1: qoff - 1 + m].copy()
    ctx.load_series(s)
    for eps in (0.5, 0.76, 3.5, 50.0):
        for ea in (1, 0):
            ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ea)
            got = kvmatch.QueryEngine(ctx).verify((L, R), 1, q, eps)
            compare(got, oracle.rsm_ed(s, q, L, R, 0, eps), tag=f"head eps {eps} abandon {ea}")
    ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 1)
    target = int(L[5])
    w = s[target - 1: target - 1 + m]
    d2 = _seq_sum((w - q) * (w - q))
    if d2 > 0:
        eps_in, eps_out = _eps_for_d2(d2)
        a = kvmatch.QueryEngine(ctx).verify((L, R), 1, q, eps_in)
        b = kvmatch.QueryEngine(ctx).verify((L, R), 1, q, eps_out)
        compare(a, oracle.rsm_ed(s, q, L, R, 0, eps_in), tag="head knife in")
        compare(b, oracle.rsm_ed(s, q, L, R, 0, eps_out), tag="head knife out")
        assert target in a.offsets.tolist() and target not in b.offsets.tolist()


def _ref_running_stats(s, run_start, start, m):
    """mean/std of the window `start` as the reference's running sums give them when the interval begins at
    `run_start` (K/NormQueryEngine.java:487-509,523-524), replayed in Python."""
    ex = ex2 = 0.0
    b = run_start - 1
    last = start - 1 + m - 1
    for i in range(b, last + 1):
        d = float(s[i])
        ex += d
        ex2 += d * d
        if i - b >= m - 1 and i < last:
            o = float(s[i - m + 1])
            ex -= o
            ex2 -= o * o
    mean = ex / m
    return mean, float(np.sqrt(ex2 / m - mean * mean))


def test_knife_edge_cnsm_alpha_beta(world):
    kvmatch, datagen, series, ctx = world
    s = series[:50_000].copy()
    m = 300
    qoff = 12_345
    q = datagen.plant_copies(s, qoff, m, [30_000], [1.25], [0.8], 0.01, seed=5)
    ctx.load_series(s)
    # one interval so that the running sums restart at a known place; the target window sits 40 into it
    run_start, target = 29_960, 30_000
    L = np.array([qoff, run_start], dtype=np.int32)
    R = np.array([qoff + 3, run_start + 80], dtype=np.int32)
    mean, std = _ref_running_stats(s, run_start, target, m)
    meanQ, stdQ = oracle.query_stats(q)
    beta_on = abs(mean - meanQ)
    ratio = std / stdQ
    assert ratio > 1.0
    for alpha, beta in ((2.0, beta_on), (2.0, float(np.nextafter(beta_on, 0.0))),
                        (ratio, 10.0), (float(np.nextafter(ratio, 0.0)), 10.0)):
        got = kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, 50.0, alpha, beta)
        want = oracle.cnsm_ed(s, q, L, R, 0, 50.0, alpha, beta)
        compare(got, want, tag=f"knife alpha={alpha!r} beta={beta!r}")
        inside = (abs(mean - meanQ) <= beta) and (std / stdQ <= alpha) and (std / stdQ >= 1.0 / alpha)
        assert (target in got.offsets.tolist()) == inside
        gd = kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, 50.0, 9, alpha, beta)
        compare(gd, oracle.cnsm_dtw(s, q, L, R, 0, 50.0, 9, alpha, beta), tag="knife dtw")
        assert (target in gd.offsets.tolist()) == inside
    assert got.stats[6] >= 1  # KVM_STAT_RECHECKED: the exact pass ran


def test_knife_edge_dtw_epsilon(world):
    kvmatch, datagen, series, ctx = world
    s = series[:40_000].copy()
    m, rho = 200, 11
    qoff = 9_000
    q = datagen.plant_copies(s, qoff, m, [25_000], [1.0], [0.0], 0.05, seed=6)
    ctx.load_series(s)
    L, R = datagen.gen_intervals(len(s), m, 6_000, 24.0, must_cover=[qoff, 25_000], seed=2)
    d2 = oracle.dtw_full(s[25_000 - 1: 25_000 - 1 + m], q, rho)
    eps_in, eps_out = _eps_for_d2(d2)
    a = kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, eps_in, rho)
    b = kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, eps_out, rho)
    compare(a, oracle.rsm_dtw(s, q, L, R, 0, eps_in, rho), tag="dtw knife in")
    compare(b, oracle.rsm_dtw(s, q, L, R, 0, eps_out, rho), tag="dtw knife out")
    assert 25_000 in a.offsets.tolist() and 25_000 not in b.offsets.tolist()


def test_degenerate_windows_constant_segments(world):
    """sigma == 0 (constant stretch) and eps == 0: NaN / division-by-zero paths must agree with the reference."""
    kvmatch, datagen, series, ctx = world
    s = series[:30_000].copy()
    s[10_000:12_000] = 3.25      # constant: variance exactly 0 -> std 0 (or NaN) -> never passes the filter
    m = 256
    qoff = sortable_query(s, m, 20_000)
    q = s[qoff - 1: qoff - 1 + m].copy()
    ctx.load_series(s)
    L = np.array([9_900, qoff - 5], dtype=np.int32)
    R = np.array([11_950, qoff + 5], dtype=np.int32)
    compare(kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, 5.0, 1.5, 100.0),
            oracle.cnsm_ed(s, q, L, R, 0, 5.0, 1.5, 100.0), tag="const cnsm_ed")
    compare(kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, 5.0, 8, 1.5, 100.0),
            oracle.cnsm_dtw(s, q, L, R, 0, 5.0, 8, 1.5, 100.0), tag="const cnsm_dtw")
    a = kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 0.0)       # eps = 0: exact-match search
    compare(a, oracle.rsm_ed(s, q, L, R, 0, 0.0), tag="eps0")
    assert a.offsets.tolist() == [qoff] and a.distances[0] == 0.0
    qc = s[10_100 - 1: 10_100 - 1 + m].copy()                     # constant query: stdQ == 0
    compare(kvmatch.NormQueryEngine(ctx).verify((L, R), 1, qc, 5.0, 1.5, 100.0),
            oracle.cnsm_ed(s, qc, L, R, 0, 5.0, 1.5, 100.0), tag="const query")
    compare(kvmatch.QueryEngine(ctx).verify((L, R), 1, qc, 1.0), oracle.rsm_ed(s, qc, L, R, 0, 1.0), tag="const rsm")


# ---------------------------------------------------------------------------------------------------------------
# Shapes at the edges of what the kernels tile for.

@pytest.mark.parametrize("m", [1, 2, 5, 31, 63, 65, 159, 160, 161, 320, 20_000])
def test_query_lengths_all_variants(world, m):
    """m below one lane group, around the 160-point chunk, and beyond one prefix-sum block (16384)."""
    kvmatch, datagen, series, ctx = world
    s = series[:120_000].copy()
    qoff = sortable_query(s, m, 50_000, step=211) if m >= 2 else 50_000
    q = datagen.plant_copies(s, qoff, m, [10_000, 90_000], [1.0, 1.0], [0.01, -0.01], 0.01, seed=m)
    ctx.load_series(s)
    L, R = datagen.gen_intervals(len(s), m, 3_000, 12.0, must_cover=[qoff, 10_000, 90_000], seed=m)
    eps = 0.5 * np.sqrt(m) + 0.2
    for k in (1, 3):
        ctx.set_option(kvmatch.OPT_ED_KERNEL, k)
        compare(kvmatch.QueryEngine(ctx).verify((L, R), 1, q, eps), oracle.rsm_ed(s, q, L, R, 0, eps), tag=f"len rsm k{k}")
        if m >= 2:
            compare(kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, eps, 2.0, 5.0),
                    oracle.cnsm_ed(s, q, L, R, 0, eps, 2.0, 5.0), tag=f"len cnsm k{k}")
    ctx.set_option(kvmatch.OPT_ED_KERNEL, 0)
    # DTW: the reference's LB_Kim reads q[0..2] and q[m-3..m-1] and double-counts below m = 6 (and every entry point
    # requires m >= 25 = WuList[0], K/NormQueryEngineDtw.java:112-162), so DTW parity is defined for m >= 25
    if 25 <= m <= 320:
        rho = max(0, m // 10)
        compare(kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, eps, rho), oracle.rsm_dtw(s, q, L, R, 0, eps, rho),
                tag="len rsm_dtw")
        if m >= 2:
            compare(kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, eps, rho, 2.0, 5.0),
                    oracle.cnsm_dtw(s, q, L, R, 0, eps, rho, 2.0, 5.0), tag="len cnsm_dtw")


@pytest.mark.parametrize("m", [65_536, 262_144])
def test_ed_guard_band_scales_with_the_query_length(built, m):
    """The ED pruner's guard band is a function of the number of terms it accumulates (kvm_ed_dot.hip, edd_dot_slack), not a
    constant that holds up to some m: queries of 2^16 and 2^18 points -- four and sixteen prefix-sum blocks, a level of 1e3 under
    the data so that the correlation form cancels -- with eps placed ON candidates' distances, one ulp and 1e-15 .. 1e-11
    relatively to either side (profiles/fuzz_knife.py's method): the pruner must hand every such candidate to the exact pass,
    whose decision is the reference's.  RSM-ED and cNSM-ED, early abandoning on and off (host-planned tile classes and window
    splits), the cross-check kernel once."""
    import kvmatch
    from kvmatch import datagen
    n = 5 * m + 20_000
    s = datagen.gen_series(n, seed=m) + 1000.0
    qoff = n // 3
    planted = [7, n - m - 40]
    q = datagen.plant_copies(s, qoff, m, planted, [1.0, 1.0], [0.004, -0.006], 0.002, seed=m + 1)
    rng = np.random.default_rng(m)
    centres = sorted([qoff] + planted + rng.integers(1, n - m, 5).tolist())
    L = np.array([max(1, c - 20) for c in centres], dtype=np.int32)
    R = np.array([min(n - m + 1, c + 25) for c in centres], dtype=np.int32)

    def around(x):
        out = [x, float(np.nextafter(x, 0.0)), float(np.nextafter(x, np.inf))]
        return out + [x * (1.0 + o) for o in (1e-15, -1e-15, 1e-13, -1e-13, 1e-11, -1e-11)]

    with kvmatch.Phase2Context(0) as ctx:
        ctx.load_series(s)
        cases = (("rsm_ed", lambda e: kvmatch.QueryEngine(ctx).verify((L, R), 1, q, e), lambda e: oracle.rsm_ed(s, q, L, R, 0, e), 1e30),
                 ("cnsm_ed", lambda e: kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, e, 2.0, 1e9),
                  lambda e: oracle.cnsm_ed(s, q, L, R, 0, e, 2.0, 1e9), 1e150))
        for name, lib, orc, wide in cases:
            o_off, o_dist, _ = orc(wide)
            assert len(o_dist) >= 100
            order = np.argsort(o_dist)
            picks = [order[1], order[2], order[len(order) // 50], order[len(order) // 8], order[len(order) // 2], order[-1]]
            for ab in (1, 0):
                ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
                for i in picks:
                    for eps in around(float(o_dist[i])):
                        compare(lib(eps), orc(eps), tag=f"{name} m={m} abandon={ab} eps at candidate {int(o_off[i])}")
            ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 1)
            ctx.set_option(kvmatch.OPT_ED_KERNEL, 1)
            compare(lib(float(o_dist[picks[2]])), orc(float(o_dist[picks[2]])), tag=f"{name} m={m} wave kernel")
            ctx.set_option(kvmatch.OPT_ED_KERNEL, 0)


@pytest.mark.parametrize("k", [5, 40])
def test_ed_guard_band_holds_against_coherent_rounding(built, k):
    """The worst case a length-m rounding bound has to cover, built on purpose: a self-match (true distance 0) whose window is
    0, 1, v, v, v, ... -- the pruner's dot product adds the same tiny product v (v - mean q) to an accumulator of ~1 four
    million times, and every one of those additions rounds the same way by the same fraction of an ulp.  With the constant
    1e-11 that rounds 2 - 5 kept around eps^2 the pruner drops this answer at m = 2^22 for these v (profiles/
    r06_guard_band_coherent_m4M.log: 22 of 60 scanned values lost; with no band at all, 25 of 60 already at m = 2^20); the band
    derived from the number of terms (kvm_ed_dot.hip, edd_dot_slack) keeps it, and the exact pass returns the reference's 0.0."""
    import kvmatch
    m = 1 << 22
    v = 1e-8 * (1.0 + k * 1.37e-3)
    s = np.full(m + 4096, v)
    qoff = 1001
    s[qoff - 1], s[qoff] = 0.0, 1.0
    q = s[qoff - 1: qoff - 1 + m].copy()
    L, R = np.array([qoff], dtype=np.int32), np.array([qoff + 2], dtype=np.int32)
    want = oracle.rsm_ed(s, q, L, R, 0, 1.4e-6)
    assert want[0].tolist() == [qoff] and want[1][0] == 0.0
    with kvmatch.Phase2Context(0) as ctx:
        ctx.load_series(s)
        for ab in (1, 0):
            ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
            compare(kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 1.4e-6), want, tag=f"coherent rounding k={k} abandon={ab}")


def test_interval_shapes(world):
    """isolated candidates, interval lengths around the 32 / 64 tile sizes, one very long interval, the last window."""
    kvmatch, datagen, series, ctx = world
    s = series[:150_000].copy()
    m = 200
    q = datagen.plant_copies(s, 70_000, m, [1, len(s) - m + 1], [1.0, 1.0], [0.0, 0.0], 0.01, seed=1)
    ctx.load_series(s)
    lens = [1, 1, 1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 5000]
    left, right, pos = [], [], 1
    for ln in lens:
        left.append(pos)
        right.append(pos + ln - 1)
        pos += ln + 7
    left += [69_990, len(s) - m - 40]
    right += [70_010, len(s) + 500]          # the last interval runs past n - m + 1 and is clipped
    L, R = np.array(left, dtype=np.int32), np.array(right, dtype=np.int32)
    for k in (1, 3):
        ctx.set_option(kvmatch.OPT_ED_KERNEL, k)
        compare(kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 3.0), oracle.rsm_ed(s, q, L, R, 0, 3.0), tag=f"shapes k{k}")
        compare(kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, 3.0, 1.5, 3.0),
                oracle.cnsm_ed(s, q, L, R, 0, 3.0, 1.5, 3.0), tag=f"shapes cnsm k{k}")
    ctx.set_option(kvmatch.OPT_ED_KERNEL, 0)
    a = kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, 3.0, 10)
    compare(a, oracle.rsm_dtw(s, q, L, R, 0, 3.0, 10), tag="shapes dtw")
    assert {1, 70_000, len(s) - m + 1} <= set(a.offsets.tolist())
    compare(kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, 3.0, 10, 1.5, 3.0),
            oracle.cnsm_dtw(s, q, L, R, 0, 3.0, 10, 1.5, 3.0), tag="shapes cnsm dtw")


def test_many_answers_and_tiny_series(world):
    """every candidate an answer (huge epsilon: exercises the > 1024-answer readback and the exact pass at scale), and a
    series shorter than every tile / block size."""
    kvmatch, datagen, series, ctx = world
    s = series[:30_000].copy()
    m = 100
    q = s[5_000:5_000 + m].copy()
    ctx.load_series(s)
    L, R = datagen.gen_intervals(len(s), m, 6_000, 24.0, must_cover=[5_001], seed=12)
    a = kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 1e6)
    compare(a, oracle.rsm_ed(s, q, L, R, 0, 1e6), tag="all answers rsm")
    assert len(a) == a.cnt_candidate > 1024
    b = kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, 1e6, 1e6, 1e6)
    compare(b, oracle.cnsm_ed(s, q, L, R, 0, 1e6, 1e6, 1e6), tag="all answers cnsm")
    c = kvmatch.NormQueryEngineDtw(ctx).verify((L[:40], R[:40]), 1, q, 1e6, 5, 1e6, 1e6)
    compare(c, oracle.cnsm_dtw(s, q, L[:40], R[:40], 0, 1e6, 5, 1e6, 1e6), tag="all answers dtw")
    t = series[:300].copy()
    ctx.load_series(t)
    qq = t[100:140].copy()
    Ls, Rs = np.array([1], dtype=np.int32), np.array([261], dtype=np.int32)
    compare(kvmatch.QueryEngine(ctx).verify((Ls, Rs), 1, qq, 2.0), oracle.rsm_ed(t, qq, Ls, Rs, 0, 2.0), tag="tiny rsm")
    compare(kvmatch.NormQueryEngineDtw(ctx).verify((Ls, Rs), 1, qq, 2.0, 4, 2.0, 5.0),
            oracle.cnsm_dtw(t, qq, Ls, Rs, 0, 2.0, 4, 2.0, 5.0), tag="tiny dtw")


def test_cnsm_ed_scattered_filter_passes_retile_and_split(world):
    """cNSM-ED with a filter that passes scattered, short stretches of each run: the device-side retile (trimmed tiles,
    class A/B lists built with atomics), the device-side window-split decision and the early abandon over ranges all
    have to agree with the oracle, whatever the (nondeterministic) order of the lists."""
    kvmatch, datagen, series, ctx = world
    rng = np.random.default_rng(99)
    n = 120_000
    # noise whose level and spread change every few hundred points: mean/std of a short window jump around, so the
    # (alpha, beta) filter passes isolated windows and short stretches
    seg = rng.integers(150, 600, 400)
    s = np.concatenate([rng.normal(rng.uniform(-2, 2), rng.uniform(0.3, 2.0), int(k)) for k in seg])[:n].copy()
    n = len(s)
    for m, eps, alpha, beta in [(64, 7.0, 1.04, 0.15), (200, 14.0, 1.02, 0.1), (1000, 35.0, 1.01, 0.05), (3000, 60.0, 1.005, 0.03)]:
        qoff = int(rng.integers(1, n - m))
        q = datagen.plant_copies(s, qoff, m, [int(rng.integers(1, n - m))], [1.0], [0.0], 0.01, seed=m)
        ctx.load_series(s)
        L, R = datagen.gen_intervals(n, m, 30_000, 300.0, must_cover=[qoff], seed=m)
        want = oracle.cnsm_ed(s, q, L, R, 0, eps, alpha, beta)
        for ab in (1, 0):
            ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
            got = kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, eps, alpha, beta)
            compare(got, want, tag=f"scattered m={m} abandon={ab}")
            assert 0 < got.stats[1] < got.stats[0] / 2  # the filter really is selective here
    ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 1)


def test_window_means_for_index_build(world):
    """sliding means for w in WuList = {25, 50, 100, 200, 400} (K/IndexBuilder.java) from the prefix
    sums; compared with exactly-rounded sums (math.fsum) and with the reference's running-sum loop restated in numpy."""
    import math
    kvmatch, datagen, series, ctx = world
    s = series[:70_000].copy()
    ctx.load_series(s)
    for w in (25, 50, 100, 200, 400):
        got = ctx.window_means(w)
        assert got.shape == (len(s) - w + 1,)
        cs = np.concatenate(([0.0], np.cumsum(s)))
        approx = (cs[w:] - cs[:-w]) / w
        assert np.allclose(got, approx, rtol=1e-9, atol=1e-9)
        for i in (0, 1, 16_383, 16_384, 16_385, 40_000, len(s) - w):
            exact = math.fsum(s[i:i + w]) / w
            assert abs(got[i] - exact) <= 1e-13 * max(1.0, abs(exact)), (w, i)
    sub = ctx.window_means(100, first=500, count=10)
    assert np.array_equal(sub, ctx.window_means(100)[499:509])
    with pytest.raises(kvmatch.KvMatchIllegalArgument):
        ctx.window_means(100, first=len(s) - 50, count=10)


def test_index_rows_bit_identical_to_reference_loop(world):
    """index build step 1 (K/IndexBuilder.java:186-300) on the GPU -- row keys and interval
    boundaries identical to the restated reference loop, bit for bit, for every WuList window length, on series shorter
    than, equal to and longer than the reference's EPOCH = 100000 buffers (running-sum restarts), and on a series with
    long constant stretches (intervals cut at 255 positions)."""
    kvmatch, datagen, series, ctx = world
    rng = np.random.default_rng(77)
    flat = np.repeat(rng.uniform(-3, 3, 400), 700)[:250_123]
    cases = [("short", series[:3_000].copy()), ("one buffer", series[:100_000].copy()),
             ("buffer+1", series[:100_001].copy()), ("walk", series[:350_017].copy()), ("flat", flat)]
    for tag, s in cases:
        ctx.load_series(s)
        for w in (25, 50, 100, 200, 400):
            keys, first, last = ctx.index_rows(w)
            wk, wf, wl = oracle.index_rows(s, w)
            assert len(keys) == len(wk), (tag, w, len(keys), len(wk))
            assert np.array_equal(first, wf) and np.array_equal(last, wl), (tag, w)
            assert np.array_equal(keys.view(np.int64), wk.view(np.int64)), (tag, w)
    s = series[:5_000].copy()
    ctx.load_series(s)
    for w in (1, 4_999, 5_000):
        keys, first, last = ctx.index_rows(w)
        wk, wf, wl = oracle.index_rows(s, w)
        assert np.array_equal(first, wf) and np.array_equal(last, wl) and np.array_equal(keys.view(np.int64), wk.view(np.int64)), w
    with pytest.raises(kvmatch.KvMatchIllegalArgument):
        ctx.index_rows(5_001)
    with pytest.raises(kvmatch.KvMatchIllegalArgument):
        ctx.index_rows(0)


def test_ucr_full_scan_executors(world):
    """the index-free full scans (U/UcrEdQueryExecutor.java, U/UcrDtwQueryExecutor.java) as a GPU
    mode, against the restated executors -- offsets, order and distances bit-identical; series longer than two of the
    DTW executor's EPOCH buffers, answers planted across a buffer boundary.  The segmented ED mode restarts the running
    sums per segment (engine semantics): bit-identical to the engine restatement on that list, and the same answers as
    the single-chain executor up to the last bits of the distances."""
    kvmatch, datagen, series, ctx = world
    from kvmatch.ucr import segments, EPOCH
    n, m = 230_000, 256
    s = series[:n].copy()
    q = datagen.plant_copies(s, sortable_query(s, m, 20_000), m, [99_400, 99_900, 150_000, 229_000], [1.0, 1.0, 1.3, 0.7], [0.0, 0.0, 0.4, -0.5], 0.03)
    ctx.load_series(s)
    eps, alpha, beta, rho = 2.5, 1.5, 1.0, 12
    ex = kvmatch.UcrEdQueryExecutor(ctx, n, m, q, eps, alpha, beta)
    wo, wd, wst = oracle.ucr_ed(s, q, eps, alpha, beta)
    assert ex.run() == len(wo) >= 4
    assert np.array_equal(ex.answers.offsets, wo) and np.array_equal(ex.answers.distances, wd)
    assert ex.answers.stats[1] == wst["pass_filter"]
    seg = kvmatch.UcrEdQueryExecutor(ctx, n, m, q, eps, alpha, beta, segment=10_000)
    assert seg.run() == len(wo)
    L, R = segments(1, n - m + 1, 10_000)
    compare(seg.answers, oracle.cnsm_ed(s, q, L, R, 0, eps, alpha, beta), tag="segmented ucr_ed")
    assert sorted(seg.answers.offsets.tolist()) == sorted(wo.tolist())
    od = dict(zip(wo.tolist(), wd.tolist()))
    assert rel_close(seg.answers.distances, np.array([od[o] for o in seg.answers.offsets.tolist()]), abs_floor=1e-9)
    dx = kvmatch.UcrDtwQueryExecutor(ctx, n, m, q, eps, rho, alpha, beta)
    wo, wd, wst = oracle.ucr_dtw(s, q, eps, rho, alpha, beta)
    assert dx.run() == len(wo) >= 4
    assert np.array_equal(dx.answers.offsets, wo) and np.array_equal(dx.answers.distances, wd)
    assert dx.answers.stats[1] == wst["pass_filter"]
    # N < series length: only the first N points are scanned
    half = kvmatch.UcrDtwQueryExecutor(ctx, 120_000, m, q, eps, rho, alpha, beta)
    wo, wd, _ = oracle.ucr_dtw(s[:120_000], q, eps, rho, alpha, beta)
    assert half.run() == len(wo) and np.array_equal(half.answers.offsets, wo) and np.array_equal(half.answers.distances, wd)
    with pytest.raises(kvmatch.KvMatchIllegalArgument):
        kvmatch.UcrEdQueryExecutor(ctx, n + 1, m, q, eps, alpha, beta)


def test_long_queries_and_wide_bands(world):
    """Bands wider than the register DP holds (rho > 511) run the generic anti-diagonal kernel: band rows + window in LDS
    while they fit (m + 2 rho <= ~20 000 points), else in a global scratch; windows longer than a prefix-sum block
    (16 384 points) span three blocks.  All against the oracle, bit for bit."""
    kvmatch, datagen, series, ctx = world
    n = 120_000
    s = series[:n].copy()
    for m, rho in ((3_000, 600), (20_000, 1_000), (40_000, 520)):
        qoff = sortable_query(s, m, 31_337 if m < 30_000 else 20_011, step=101)   # (its window ends in front of the copy at 70 001)
        q = datagen.plant_copies(s, qoff, m, [70_001], [1.05], [0.2], 0.01, seed=m)
        ctx.load_series(s)
        L = np.array([qoff - 2, 50_000, 70_000], dtype=np.int32)
        R = np.array([qoff + 2, 50_003, 70_002], dtype=np.int32)
        eps = 0.05 * np.sqrt(m)
        compare(kvmatch.QueryEngine(ctx).verify((L, R), 1, q, eps), oracle.rsm_ed(s, q, L, R, 0, eps), tag=f"rsm_ed m={m}")
        compare(kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, eps, 1.5, 5.0),
                oracle.cnsm_ed(s, q, L, R, 0, eps, 1.5, 5.0), tag=f"cnsm_ed m={m}")
        got = kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, eps, rho)
        compare(got, oracle.rsm_dtw(s, q, L, R, 0, eps, rho), tag=f"rsm_dtw m={m} rho={rho}")
        assert qoff in got.offsets.tolist()
        compare(kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, eps, rho, 1.5, 5.0),
                oracle.cnsm_dtw(s, q, L, R, 0, eps, rho, 1.5, 5.0), tag=f"cnsm_dtw m={m} rho={rho}")
    ctx.load_series(series)


def test_index_rows_grouped(world):
    """Rows of step 1 (one per distinct key, descending key order, intervals in position order) built on the GPU
    (stable radix sort on the Double.compareTo-ordered bit pattern) against the restatement, bit for bit."""
    kvmatch, datagen, series, ctx = world
    rng = np.random.default_rng(78)
    mixed = series[:200_000].copy()
    mixed[5_000:6_000] = -0.0
    mixed[50_000:50_003] = np.nan
    flat = np.repeat(rng.uniform(-3, 3, 300), 700)[:180_123]
    for tag, s in (("walk", series[:250_017].copy()), ("mixed", mixed), ("flat", flat)):
        ctx.load_series(s)
        for w in (25, 100, 400):
            got = ctx.index_rows(w, grouped=True)
            want = oracle.index_rows_grouped(s, w)
            for a, b, name in zip(got, want, ("keys", "first", "last", "group_keys", "group_offsets")):
                if a.dtype == np.float64:
                    assert np.array_equal(a.view(np.int64), b.view(np.int64)), (tag, w, name)
                else:
                    assert np.array_equal(a, b), (tag, w, name)
            assert len(ctx.index_rows(w)[0]) == len(got[0])
    ctx.load_series(series)


def test_cpp_host_mirror_cli(world):
    """The C++ host mirror (kv-match_amd/host/kvmatch_engines.hpp) through its CLI driver: same seeded inputs as the Python
    mirror, same answers -- engines, full-scan executors, index rows (plain and grouped), phase-2 time model."""
    import os, re, subprocess
    kvmatch, datagen, series, ctx = world
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kv-match_amd", "host", "kvm_cli")
    n, m, eps = 400_000, 512, 4.0
    out = subprocess.run([cli, str(n), str(m), str(eps)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = {l.split()[0] + (" " + l.split()[1] if l.split()[1] in ("ED", "DTW") else ""): l for l in out.stdout.splitlines() if l.strip()}
    s = datagen.gen_series(n)
    qoff = n // 3 + 17
    q = s[qoff - 1: qoff - 1 + m].copy()
    L, R = datagen.gen_intervals(n, m, n // 100, 64.0, must_cover=[qoff], seed=1)
    c2 = kvmatch.Phase2Context(0)
    c2.load_series(s)
    rho = c2._lib.kvm_resolve_rho(0.05, m)
    want = {"RSM-ED": kvmatch.QueryEngine(c2).verify((L, R), 1, q, eps),
            "RSM-DTW": kvmatch.QueryEngineDtw(c2).verify((L, R), 1, q, eps, 0.05),
            "cNSM-ED": kvmatch.NormQueryEngine(c2).verify((L, R), 1, q, eps, 1.5, 5.0),
            "cNSM-DTW": kvmatch.NormQueryEngineDtw(c2).verify((L, R), 1, q, eps, 0.05, 1.5, 5.0)}
    ue = kvmatch.UcrEdQueryExecutor(c2, n, m, q, eps, 1.5, 5.0, segment=50_000); ue.run()
    ud = kvmatch.UcrDtwQueryExecutor(c2, n, m, q, eps, rho, 1.5, 5.0); ud.run()
    want["UCR-ED"], want["UCR-DTW"] = ue.answers, ud.answers
    for name, a in want.items():
        line = next(l for l in out.stdout.splitlines() if l.startswith(name))
        mo = re.search(r"Best: (-?\d+), distance: (\S+) .*#answers: (\d+)", line)
        assert mo, line
        assert int(mo.group(1)) == int(a.offsets[0]) and int(mo.group(3)) == len(a), (name, line)
        assert abs(float(mo.group(2)) - a.distances[0]) <= 1e-5 * max(1.0, a.distances[0]), (name, line)
    # the C++ QueryStream (kvm_run_jobs: three contexts, native host threads): first and last job of a six-job stream
    for tag, name in (("stream[0]", "RSM-ED"), ("stream[5]", "cNSM-ED")):
        line = next(l for l in out.stdout.splitlines() if l.startswith(tag))
        mo = re.search(r"Best: (-?\d+), distance: (\S+) .*#answers: (\d+)", line)
        assert mo and int(mo.group(1)) == int(want[name].offsets[0]) and int(mo.group(3)) == len(want[name]), line
    keys, first, last, gk, go = c2.index_rows(50, grouped=True)
    idx = next(l for l in out.stdout.splitlines() if l.startswith("index w=50"))
    nums = [int(x) for x in re.findall(r"\d+", idx)]
    assert nums[1] == len(keys) and nums[2] == n - 49 and nums[3] == len(gk) and nums[4] == len(keys) and "keys descending" in idx
    t2 = float(re.search(r"([0-9.]+) ms", next(l for l in out.stdout.splitlines() if l.startswith("T_2 model"))).group(1))
    assert abs(t2 - kvmatch.phase2_time_estimate_ms(0, len(L), n // 100, m)) < 1e-3
    c2.close()


@pytest.mark.lenient_sort   # (60 random queries, noise included: on a good part of them the reference's sort throws, and which
#                             ones is the subject of test_cnsm_dtw_sort_violation_...; here the switch carries on, on both sides)
def test_randomised_sweep_all_variants(world):
    """Seeded random sweep over series shape, query length, thresholds and interval layouts: 240 verifications, each
    compared bit for bit with the oracle."""
    kvmatch, datagen, series, ctx = world
    rng = np.random.default_rng(20260101)
    cases = died = 0
    for trial in range(60):
        n = int(rng.integers(3_000, 60_000))
        kind = trial % 4
        if kind == 0:
            s = series[int(rng.integers(0, 200_000)):][:n].copy()
        elif kind == 1:
            s = np.cumsum(rng.uniform(-1, 1, n)) + rng.uniform(-2000, 2000)     # long random walk, large offset
        elif kind == 2:
            s = rng.normal(rng.uniform(-5, 5), rng.uniform(0.01, 2), n)           # noise
            s[n // 3: n // 3 + 500] = s[n // 3]                                   # with a constant stretch
        else:
            s = 5 * np.sin(np.arange(n) * rng.uniform(0.001, 0.3)) + rng.normal(0, 0.1, n)
        m = int(rng.choice([25, 32, 47, 64, 100, 128, 255, 300, 512, 1000]))
        if m * 3 > n:
            m = 25
        qoff = int(rng.integers(1, n - m + 1))
        planted = [int(p) for p in rng.integers(1, n - m + 1, 2)]
        q = datagen.plant_copies(s, qoff, m, planted, [1.0, float(rng.uniform(0.8, 1.3))],
                                 [0.0, float(rng.uniform(-1, 1))], float(rng.uniform(0.001, 0.1)), seed=trial)
        ctx.load_series(s)
        mean_len = float(rng.choice([1.0, 3.0, 16.0, 64.0, 300.0]))
        L, R = datagen.gen_intervals(n, m, int(rng.integers(200, 4000)), mean_len, must_cover=[qoff] + planted,
                                     seed=trial)
        last_segment = int(rng.integers(1, 4))
        shift = (last_segment - 1) * 25
        L2, R2 = (L + shift).astype(np.int32), (R + shift).astype(np.int32)
        eps = float(rng.choice([0.0, 0.05, 0.3, 1.0, 3.0])) * np.sqrt(m) * 0.2 + float(rng.uniform(0, 0.5))
        alpha = float(rng.choice([1.0, 1.05, 1.5, 3.0, 100.0]))
        beta = float(rng.choice([0.0, 0.1, 1.0, 10.0, 1e6]))
        rho = int(rng.choice([0, 1, 2, m // 20, m // 10, m // 3]))
        rho_resolved = oracle.resolve_rho(rho, m)  # the reference's prompt rule: values <= 1 are a fraction of m
        tag = f"trial {trial} n={n} m={m} eps={eps:.3f} a={alpha} b={beta} rho={rho} ml={mean_len}"
        compare(kvmatch.QueryEngine(ctx).verify((L2, R2), last_segment, q, eps),
                oracle.rsm_ed(s, q, L2, R2, shift, eps), tag="rsm_ed " + tag)
        compare(kvmatch.NormQueryEngine(ctx).verify((L2, R2), last_segment, q, eps, alpha, beta),
                oracle.cnsm_ed(s, q, L2, R2, shift, eps, alpha, beta), tag="cnsm_ed " + tag)
        # (a float rho goes through the prompt rule, an int is points: alternate the two spellings)
        rho_arg = float(rho) if trial % 2 else int(rho_resolved)
        if rho_resolved >= m:
            # (rho = 1 typed at the reference's prompt is "100 % of the length": lowerUpperLemire then indexes below its arrays and
            # the query dies, K/utils/DtwUtils.java:81-90 -- on both sides here)
            for f in (lambda: kvmatch.QueryEngineDtw(ctx).verify((L2, R2), last_segment, q, eps, rho_arg),
                      lambda: kvmatch.NormQueryEngineDtw(ctx).verify((L2, R2), last_segment, q, eps, rho_arg, alpha, beta)):
                with pytest.raises(kvmatch.KvMatchIndexOutOfBounds):
                    f()
            for f in (lambda: oracle.rsm_dtw(s, q, L2, R2, shift, eps, rho_resolved),
                      lambda: oracle.cnsm_dtw(s, q, L2, R2, shift, eps, rho_resolved, alpha, beta)):
                with pytest.raises(oracle.OracleError) as ei:
                    f()
                assert ei.value.code == -8
            died += 1
            cases += 4
            continue
        compare(kvmatch.QueryEngineDtw(ctx).verify((L2, R2), last_segment, q, eps, rho_arg),
                oracle.rsm_dtw(s, q, L2, R2, shift, eps, rho_resolved), tag="rsm_dtw " + tag)
        compare(kvmatch.NormQueryEngineDtw(ctx).verify((L2, R2), last_segment, q, eps, rho_arg, alpha, beta),
                oracle.cnsm_dtw(s, q, L2, R2, shift, eps, rho_resolved, alpha, beta), tag="cnsm_dtw " + tag)
        cases += 4
    assert cases == 240 and died >= 1     # (some trial has rho = 1.0 -> the whole length)


@pytest.mark.parametrize("m,rho", [(3000, 1100), (9000, 2000)])
def test_wide_bands_envelope_paths(world, m, rho):
    """rho > 1024 (1024-point envelope tiles with > 64 KiB of LDS) and the LDS form of the DP (rho > 511)."""
    kvmatch, datagen, series, ctx = world
    s = series[:60_000].copy()
    qoff = sortable_query(s, m, 20_000, step=53)
    q = datagen.plant_copies(s, qoff, m, [40_000], [1.0], [0.0], 0.02, seed=2)
    ctx.load_series(s)
    L, R = datagen.gen_intervals(len(s), m, 300, 8.0, must_cover=[qoff, 40_000], seed=3)
    eps = 0.08 * np.sqrt(m)
    compare(kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, eps, rho), oracle.rsm_dtw(s, q, L, R, 0, eps, rho),
            tag=f"wide band rsm {rho}")
    compare(kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, eps, rho, 1.5, 3.0),
            oracle.cnsm_dtw(s, q, L, R, 0, eps, rho, 1.5, 3.0), tag=f"wide band cnsm {rho}")


@pytest.mark.strict_sort
def test_cnsm_dtw_sort_violation_like_the_reference_and_the_lenient_option(world):
    """NormQueryEngineDtw's Arrays.sort throws "Comparison method violates its general contract!" on some queries
    (K/NormQueryEngineDtw.java:475-478 under JDK 8's TimSort): the library then fails the same way, before touching
    the candidates; OPT_SORT_VIOLATION = 1 answers anyway -- the set brute force gives (tests/bruteforce.py)."""
    import bruteforce as bf
    kvmatch, datagen, series, ctx = world
    rng = np.random.default_rng(99)
    m, rho, eps, alpha, beta = 64, 6, 5.0, 1.5, 5.0
    for _ in range(100):
        q = rng.normal(size=m)
        try:
            kvmatch.cnsm_dtw_order(q)
        except kvmatch.KvMatchIllegalArgument:
            break
    else:
        raise AssertionError("no violating query found")
    s = rng.normal(size=4000)
    for pos in (500, 2000, 3100):
        s[pos:pos + m] = q * rng.uniform(0.8, 1.2) + rng.uniform(-1, 1) + rng.normal(size=m) * 0.05
    L, R = np.array([1], dtype=np.int32), np.array([len(s) - m + 1], dtype=np.int32)
    ctx.load_series(s)
    eng = kvmatch.NormQueryEngineDtw(ctx)
    with pytest.raises(oracle.OracleError):
        oracle.cnsm_dtw(s, q, L, R, 0, eps, rho, alpha, beta)
    with pytest.raises(kvmatch.KvMatchIllegalArgument, match="Comparison method violates its general contract!"):
        eng.verify((L, R), 1, q, eps, rho, alpha, beta)
    # zero candidates: the reference still sorts first and still throws
    with pytest.raises(kvmatch.KvMatchIllegalArgument, match="general contract"):
        eng.verify((L[:0], R[:0]), 1, q, eps, rho, alpha, beta)
    # the context stays usable, and the other engines answer the same query
    compare(kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, eps, rho), oracle.rsm_dtw(s, q, L, R, 0, eps, rho), tag="rsm after violation")
    compare(kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, eps, alpha, beta), oracle.cnsm_ed(s, q, L, R, 0, eps, alpha, beta),
            tag="cnsm_ed after violation")
    ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 1)
    try:
        got = eng.verify((L, R), 1, q, eps, rho, alpha, beta)
    finally:
        ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 0)
    mq, sq = bf.znorm_stats(q)
    zq = (q - mq) / sq
    want = {}
    for st in range(1, len(s) - m + 2):
        w = s[st - 1: st - 1 + m]
        mean, std = bf.znorm_stats(w)
        if abs(mean - mq) <= beta and std / sq <= alpha and std / sq >= 1 / alpha:
            v = oracle.dtw_full((w - mean) / std, zq, rho)
            if v <= eps * eps:
                want[st] = np.sqrt(v)
    assert sorted(got.offsets.tolist()) == sorted(want) and len(want) >= 3
    assert all(abs(want[o] - d) <= 1e-6 * max(1.0, d) for o, d in zip(got.offsets.tolist(), got.distances.tolist()))
    assert all(p + 1 in want for p in (500, 2000, 3100))
    # a query the sort accepts is unaffected by the option
    q2 = s[1000:1000 + m].cumsum()
    q2 = np.sin(np.arange(m) * 0.1) * 2 + 0.01 * q2
    kvmatch.cnsm_dtw_order(q2)
    a = eng.verify((L, R), 1, q2, eps, rho, alpha, beta)
    ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 1)
    try:
        b = eng.verify((L, R), 1, q2, eps, rho, alpha, beta)
    finally:
        ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 0)
    assert np.array_equal(a.offsets, b.offsets) and np.array_equal(a.distances, b.distances)
    compare(a, oracle.cnsm_dtw(s, q2, L, R, 0, eps, rho, alpha, beta), tag="accepted query")


def test_ed_pruner_tile_classes_every_remainder_length(world):
    """The correlation-form ED pruner cuts a run into 64-candidate class-A tiles and one remainder tile of class A (33..63
    candidates), B (17..32) or C (1..16, two unrelated tiles per row, half-length chunks): runs of every length 1..140
    with a planted copy of the query at the first, the last or a middle offset of the run (in turn), both abandon modes,
    both ED engines, bit for bit against the oracle (class C only exists with early abandoning off, RSM-ED)."""
    kvmatch, datagen, series, ctx = world
    m = 300                                    # several chunks of every class: 128 / 160 / 80 points
    s = series.copy()
    rng = np.random.default_rng(123)
    q = s[5000:5000 + m].copy()
    L, R, pos = [], [], 10_000
    lengths = list(range(1, 141)) + [16, 16, 16, 17, 32, 33, 48, 49, 64, 65, 80, 81, 96, 97]
    planted = []
    for i, ln in enumerate(lengths):
        L.append(pos)
        R.append(pos + ln - 1)
        off = (pos, pos + ln - 1, pos + ln // 2)[i % 3]      # (one copy per run: copies inside one run would overlap)
        s[off - 1: off - 1 + m] = q + rng.normal(size=m) * 0.01
        planted.append(off)
        pos += ln + m + int(rng.integers(2, 40))
    assert pos + m < len(s)
    L, R = np.array(L, dtype=np.int32), np.array(R, dtype=np.int32)
    ctx.load_series(s)
    try:
        for ab in (0, 1):
            ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
            for eps in (0.5, 3.0, 40.0):
                got = kvmatch.QueryEngine(ctx).verify((L, R), 1, q, eps)
                compare(got, oracle.rsm_ed(s, q, L, R, 0, eps), tag=f"classes rsm ab={ab} eps={eps}")
                got = kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, eps * 0.3, 1.5, 3.0)
                compare(got, oracle.cnsm_ed(s, q, L, R, 0, eps * 0.3, 1.5, 3.0), tag=f"classes cnsm ab={ab} eps={eps}")
        ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0)
        got = kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 3.0)
        assert set(planted) <= set(got.offsets.tolist())      # every run's copy is an answer: every tile class held answers
    finally:
        ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 1)
        ctx.load_series(series)


def test_ed_pruner_every_window_split_and_awkward_window_lengths(world):
    """The launcher shares a unit between 1, 2 or 4 waves (each takes a range of 128-point chunks, part 0 finishes the
    candidates): every split pair forced through KVM_OPT_ED_FORCE, window lengths that leave 1..127 points for the last chunk, fewer
    chunks than parts (m = 200 split four ways: two parts have no chunk), runs of all three tile classes with a planted copy
    each; RSM-ED with every point evaluated (the device-built lists of cNSM-ED and of early abandoning split by themselves),
    bit for bit against the oracle."""
    kvmatch, datagen, series, ctx = world
    rng = np.random.default_rng(321)
    lengths = [1, 7, 16, 17, 31, 32, 33, 50, 64, 65, 100, 128, 129, 200, 16, 16, 5, 9]
    try:
        ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0)
        for m in (200, 257, 1000, 1111, 2049):
            s = series.copy()
            q = s[5000:5000 + m].copy()
            L, R, pos = [], [], 12_000
            for i, ln in enumerate(lengths):
                L.append(pos)
                R.append(pos + ln - 1)
                off = (pos, pos + ln - 1, pos + ln // 2)[i % 3]
                s[off - 1: off - 1 + m] = q + rng.normal(size=m) * 0.01
                pos += ln + m + int(rng.integers(2, 40))
            assert pos + m < len(s)
            L, R = np.array(L, dtype=np.int32), np.array(R, dtype=np.int32)
            ctx.load_series(s)
            want = oracle.rsm_ed(s, q, L, R, 0, 2.0)
            assert len(want[0]) >= len(lengths)
            for fa in (1, 2, 4):
                for fb in (1, 2, 4):
                    ctx.set_option(kvmatch.OPT_ED_FORCE, kvmatch.ed_force(split_a=fa, split_b=fb))
                    compare(kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 2.0), want, tag=f"split {fa},{fb} m={m}")
                    sh = kvmatch.ed_shape(ctx.get_option(kvmatch.OPT_ED_LAST_SHAPE))
                    assert (sh["split_a"], sh["split_b"]) == (fa, fb)
    finally:
        ctx.set_option(kvmatch.OPT_ED_FORCE, 0)
        ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 1)
        ctx.load_series(series)


def test_ed_pruner_class_c_on_a_long_list_of_short_runs(built):
    """Class-C tiles (remainders of 1..16 candidates, two per row) are only used when the list needs more than one round of
    the chip's wave slots (kvm_verify.hip: hybrid_mode): 90 000 runs of 1..48 candidates, early abandoning off -> several
    thousand units of each class, with planted copies of the query inside tiles of every class; bit for bit against
    the oracle, and the same answers with early abandoning on (device-built lists, classes A and B only)."""
    import kvmatch
    from kvmatch import datagen
    n, m = 3_000_000, 300
    s = datagen.gen_series(n, seed=99)
    rng = np.random.default_rng(99)
    q = s[1_000_000:1_000_000 + m].copy()
    L, R, pos = [], [], 1
    while pos + 48 + m < n:
        ln = int(rng.integers(1, 49))
        L.append(pos)
        R.append(pos + ln - 1)
        pos += ln + int(rng.integers(2, 13))
    L, R = np.array(L, dtype=np.int32), np.array(R, dtype=np.int32)
    assert len(L) > 80_000
    planted = []
    for i in rng.choice(len(L), 300, replace=False):
        off = int(rng.integers(L[i], R[i] + 1))
        if planted and min(abs(off - p) for p in planted) < 2 * m:
            continue
        s[off - 1: off - 1 + m] = q + rng.normal(size=m) * 0.01
        planted.append(off)
    lens = (R - L + 1)
    assert (lens <= 16).sum() > 20_000 and ((lens > 16) & (lens <= 32)).sum() > 20_000 and (lens > 32).sum() > 20_000
    with kvmatch.Phase2Context(0) as ctx:
        ctx.load_series(s)
        want = oracle.rsm_ed(s, q, L, R, 0, 2.0)
        for ab in (0, 1):
            ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
            got = kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 2.0)
            compare(got, want, tag=f"long list of short runs ab={ab}")
        assert set(planted) <= set(got.offsets.tolist()) and len(planted) > 100
        ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0)
        got = kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 30.0)       # many survivors of the pruner in every class
        compare(got, oracle.rsm_ed(s, q, L, R, 0, 30.0), tag="long list, wide eps")


def _long_list_of_short_runs(k, m, seed):
    from kvmatch import datagen
    rng = np.random.default_rng(seed)
    lens = rng.integers(1, 21, k)
    gaps = rng.integers(2, 9, k)
    L = (1 + np.cumsum(lens + gaps) - lens - gaps).astype(np.int64) + 1
    R = L + lens - 1
    n = int(R[-1]) + 4 * m
    s = datagen.gen_series(n, seed=seed)
    qo = sortable_query(s, m, n // 3 + 1)
    q = s[qo - 1: qo - 1 + m].copy()
    L, R = L.astype(np.int32), R.astype(np.int32)
    for i in rng.choice(len(L), 200, replace=False):                      # answers in every part of the list
        off = int(rng.integers(L[i], R[i] + 1))
        noise = rng.normal(size=m) * 0.01 * min(1.0, float(np.std(q)))
        s[off - 1: off - 1 + m] = 1.5 * q + 0.3 + noise if i % 2 else q + noise
    return s, q, L, R


def test_lists_of_more_than_262144_intervals(built):
    """Above 1024 blocks of 256 intervals the device-side plan scans its block totals in a kernel of its own
    (kvm_plan_scan_kernel).  That kernel once built its in-wave exclusive prefix from a four-field initialiser of the
    five-field sums (class-C tiles were added in round 2): class-C tile bases and window bases of every such list were
    wrong -- a memory fault in the tile expansion.  330 000 runs of 1..20 candidates through all four engines (early
    abandoning off and on), bit for bit against the oracle; on a series of zeros (eps = 0) millions of DTW candidates take
    the sequential tie path, in several batches of its 1-GiB slot buffer (dtw_resolve_ambiguous)."""
    import kvmatch
    m = 64
    s, q, L, R = _long_list_of_short_runs(330_000, m, 5)
    assert len(L) > 262_144 + 256
    with kvmatch.Phase2Context(0) as ctx:
        ctx.load_series(s)
        ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 1)
        for ab in (0, 1):
            ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
            got = kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 1.0)
            compare(got, oracle.rsm_ed(s, q, L, R, 0, 1.0), tag=f"330k intervals rsm_ed ab={ab}")
            assert len(got.offsets) >= 90
            got = kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, 1.0, 2.0, 1.0)
            compare(got, oracle.cnsm_ed(s, q, L, R, 0, 1.0, 2.0, 1.0), tag=f"330k intervals cnsm_ed ab={ab}")
            assert len(got.offsets) >= 90
        ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0)
        got = kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 6.0)           # many pruner survivors in every class
        compare(got, oracle.rsm_ed(s, q, L, R, 0, 6.0), tag="330k intervals rsm_ed wide")
        for eps in (1.0, 0.0):
            got = kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, eps, 3)
            compare(got, oracle.rsm_dtw(s, q, L, R, 0, eps, 3), tag=f"330k intervals rsm_dtw eps={eps}")
            got = kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, eps, 3, 2.0, 1.0, resolved=True)
            compare(got, oracle.cnsm_dtw(s, q, L, R, 0, eps, 3, 2.0, 1.0), tag=f"330k intervals cnsm_dtw eps={eps}")
        # a series of zeros with a one in a hundred, query of zeros, eps = 0: nearly every candidate's bounds land exactly
        # on the threshold -> millions of candidates through the sequential tie path, in several batches
        rng = np.random.default_rng(7)
        z = (rng.random(len(s)) < 0.01).astype(np.float64)
        ctx.load_series(z)
        got = kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, np.zeros(m), 0.0, 3)
        compare(got, oracle.rsm_dtw(z, np.zeros(m), L, R, 0, 0.0, 3), tag="330k intervals rsm_dtw ties")
        assert got.stats[kvmatch.STAT_DTW_SEQUENTIAL] > (1 << 30) // (2 * m * 8) and len(got.offsets) > 1_000_000


def test_lists_of_more_than_16_rounds_of_wave_slots(built):
    """From 16 units per wave slot on (32 768 units on this part) the pruner's launcher stops modelling window splits and
    the device-built lists are launched on a grid that never splits: 800 000 runs of 1..20 candidates are ~ 47 000 units.
    Both ED engines, early abandoning off and on."""
    import kvmatch
    m = 64
    s, q, L, R = _long_list_of_short_runs(800_000, m, 6)
    lens = R.astype(np.int64) - L + 1
    assert (lens <= 16).sum() // 16 + (lens > 16).sum() // 8 > 16 * 2048 + 4096
    with kvmatch.Phase2Context(0) as ctx:
        ctx.load_series(s)
        for ab in (0, 1):
            ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
            got = kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 1.0)
            compare(got, oracle.rsm_ed(s, q, L, R, 0, 1.0), tag=f"800k intervals rsm_ed ab={ab}")
            assert len(got.offsets) >= 90
            got = kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, 1.0, 2.0, 1.0)
            compare(got, oracle.cnsm_ed(s, q, L, R, 0, 1.0, 2.0, 1.0), tag=f"800k intervals cnsm_ed ab={ab}")
            assert len(got.offsets) >= 10


@pytest.mark.lenient_sort   # (the lists below are built around the query's own offset 50 001, a slice on which the reference's sort
#                             throws: this test is about malformed LISTS, so the switch carries on in index order on both sides)
def test_interval_lists_not_in_the_form_sort_and_merge_leaves(world):
    """The engines hand phase 2 the list sortAndMergeIntervals leaves (sorted, disjoint, non-adjacent).  Anything else a
    caller might pass: adjacent, overlapping, nested, repeated and reversed (empty) intervals go through like in the Java
    loops -- every interval scanned on its own, an offset covered twice answered twice -- bit for bit with the oracle; a
    list not sorted by `left` gives the same answers as its sorted form for the RSM engines (intervals are independent
    there) and IllegalArgument for the cNSM engines on both sides (their chunk indexing breaks in Java)."""
    kvmatch, datagen, series, ctx = world

    def compare(got, want, tag):     # (an offset may occur twice, with distances that differ in their last bits: no dict)
        assert np.array_equal(got.offsets, want[0]) and np.array_equal(got.distances, want[1]), tag
        assert got.cnt_candidate == want[2]["cnt_candidate"], tag

    m = 200
    s = series.copy()
    q = s[50_000:50_000 + m].copy()
    for p in (20_000, 120_000, 200_000):
        s[p - 1:p - 1 + m] = q + np.random.default_rng(p).normal(size=m) * 0.01
    ctx.load_series(s)
    lists = {
        "adjacent": ([100, 200, 300, 19_990, 20_001], [199, 299, 399, 20_000, 20_050]),
        "overlapping": ([100, 150, 19_990, 19_995], [200, 400, 20_010, 20_020]),
        "nested": ([100, 150, 160, 19_900, 19_990], [1000, 200, 170, 20_100, 20_010]),
        "reversed": ([500, 19_990], [400, 20_010]),
        "repeated": ([19_990, 19_990, 19_990], [20_010, 20_010, 20_010]),
        # two runs that START at the same offset but end differently: a candidate's records carry the index of its run, not
        # the run's start (looking the run up by its start took the wrong one: RSM-DTW clipped the data envelope of
        # LB_Keogh at the wrong end and lost answers near eps -- found by profiles/fuzz_lists.py with long lists)
        "same_left": ([100, 100, 19_960, 19_960, 19_960], [120, 400, 19_965, 20_100, 20_003]),
    }
    try:
        for name, (L, R) in lists.items():
            L, R = np.array(L, dtype=np.int32), np.array(R, dtype=np.int32)
            compare(kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 3.0), oracle.rsm_ed(s, q, L, R, 0, 3.0), tag=name + " rsm_ed")
            compare(kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, 1.0, 1.5, 3.0), oracle.cnsm_ed(s, q, L, R, 0, 1.0, 1.5, 3.0),
                    tag=name + " cnsm_ed")
            compare(kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, 3.0, 10), oracle.rsm_dtw(s, q, L, R, 0, 3.0, 10), tag=name + " rsm_dtw")
            compare(kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, 1.0, 10, 1.5, 3.0),
                    oracle.cnsm_dtw(s, q, L, R, 0, 1.0, 10, 1.5, 3.0), tag=name + " cnsm_dtw")
            # every window an answer: shows WHICH windows are scanned.  The cNSM engines scan an interval inside its merged
            # chunk only, and the chunk ends where its LAST interval ends (K/NormQueryEngine.java:494): in "nested" and
            # "same_left" an earlier, longer interval loses the windows that reach past it (profiles/fuzz_lists.py found the
            # library scanning them)
            compare(kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, 1e3, 1e9, 1e9), oracle.cnsm_ed(s, q, L, R, 0, 1e3, 1e9, 1e9),
                    tag=name + " cnsm_ed, every window")
            compare(kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, 1e3, 10, 1e9, 1e9),
                    oracle.cnsm_dtw(s, q, L, R, 0, 1e3, 10, 1e9, 1e9), tag=name + " cnsm_dtw, every window")
            compare(kvmatch.QueryEngine(ctx).verify((L, R), 1, q, 1e6), oracle.rsm_ed(s, q, L, R, 0, 1e6), tag=name + " rsm_ed, every window")
        # not sorted by left
        L = np.array([120_000, 100, 19_990, 50_000, 199_990], dtype=np.int32)
        R = np.array([120_100, 400, 20_010, 50_040, 200_020], dtype=np.int32)
        order = np.argsort(L, kind="stable")
        Ls, Rs = L[order], R[order]
        for eng, ofn, args in ((kvmatch.QueryEngine, oracle.rsm_ed, (3.0,)), (kvmatch.QueryEngineDtw, oracle.rsm_dtw, (3.0, 10))):
            got = eng(ctx).verify((L, R), 1, q, *args)
            want = ofn(s, q, Ls, Rs, 0, *args)                     # the sorted form of the list
            compare(got, want, tag="unsorted " + eng.__name__)
            assert {20_000, 50_001, 120_000, 200_000} <= set(got.offsets.tolist())
            raw = ofn(s, q, L, R, 0, *args)                        # the Java loops on the list as given: the same pairs
            assert sorted(zip(raw[0].tolist(), raw[1].tolist())) == sorted(zip(got.offsets.tolist(), got.distances.tolist()))
        for call, ocall in ((lambda: kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, 1.0, 1.5, 3.0),
                             lambda: oracle.cnsm_ed(s, q, L, R, 0, 1.0, 1.5, 3.0)),
                            (lambda: kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, 1.0, 10, 1.5, 3.0),
                             lambda: oracle.cnsm_dtw(s, q, L, R, 0, 1.0, 10, 1.5, 3.0))):
            with pytest.raises(kvmatch.KvMatchIllegalArgument):
                call()
            with pytest.raises(oracle.OracleError):
                ocall()
    finally:
        ctx.load_series(series)


def test_ucr_ed_executor_at_epsilon_zero(world):
    """U/UcrEdQueryExecutor.java's distance() abandons on `sum < bsf` (:93) where the engines' loops run on `dist <= bsf`: at
    Epsilon^2 == 0 it never adds a term, so the executor reports EVERY window that passes the (Alpha, Beta) test at
    distance 0.  The oracle restates that; the mirrors reproduce it (kvmatch/ucr.py)."""
    kvmatch, datagen, series, ctx = world
    n, m = 30_000, 60
    s = series[:n].copy()
    q = s[5_000:5_000 + m].copy()
    ctx.load_series(s)
    try:
        for eps, alpha, beta in ((0.0, 1.3, 1.5), (1e-200, 1e9, 1e9), (0.0, 1.0, 0.0)):
            e = kvmatch.UcrEdQueryExecutor(ctx, n, m, q, eps, alpha, beta)
            cnt = e.run()
            off, dist, st = oracle.ucr_ed(s, q, eps, alpha, beta)
            assert cnt == len(off) and np.array_equal(e.answers.offsets, off) and np.array_equal(e.answers.distances, dist)
            assert np.all(dist == 0.0) and (alpha == 1.0 or 5_001 in off.tolist())   # (alpha = 1, beta = 0: the running sums
            #                                   of the scan need not reproduce the query's own statistics to the last bit)
        assert len(oracle.ucr_ed(s, q, 0.0, 1e9, 1e9)[0]) == n - m + 1         # (the artefact: every window)
        # the engine on the same windows computes the first term and keeps the true zero-distance matches only
        L, R = np.array([1], dtype=np.int32), np.array([n - m + 1], dtype=np.int32)
        eng = kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, 0.0, 1e9, 1e9)
        compare(eng, oracle.cnsm_ed(s, q, L, R, 0, 0.0, 1e9, 1e9), tag="engine at eps 0")
        assert len(eng) < 10
    finally:
        ctx.load_series(series)


def test_ucr_ed_partial_sum_tie(world):
    """U/UcrEdQueryExecutor.java's distance() loop runs `while sum < bsf`: a partial sum (in its |zQ|-sorted order) that lands
    EXACTLY on eps^2 ends the loop and the window is reported at distance eps, whatever its real distance.  Found by
    profiles/fuzz_ucr.py (seed 261) on integer-valued data: offset 35265, real distance 10.28, reported at eps = 2.  The
    oracle restates that and, since round 3, the executor mirror reproduces it (KVM_OPT_UCR_LOOPS, kvm_ucr.hip): plain
    equality -- while the ENGINE on the same windows keeps answering with the windows whose full sum is <= eps^2."""
    kvmatch, datagen, series, ctx = world
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ucr", "ucr_ed_partial_sum_tie.npz"))
    s, q = g["s"].astype(np.float64), g["q"].astype(np.float64)
    eps, alpha, beta = float(g["eps"]), float(g["alpha"]), float(g["beta"])
    off, dist, st = oracle.ucr_ed(s, q, eps, alpha, beta)
    assert np.array_equal(off, g["oracle_off"]) and np.array_equal(dist, g["oracle_dist"])
    tie = [int(o) for o, d in zip(off, dist) if d == eps]
    assert tie == [35265]
    w = s[tie[0] - 1: tie[0] - 1 + len(q)]
    real = np.sqrt(np.sum(((w - w.mean()) / w.std() - (q - q.mean()) / q.std()) ** 2))
    assert real > 5 * eps                                                   # nowhere near a match
    ctx.load_series(s)
    try:
        for ab in (1, 0):
            ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
            e = kvmatch.UcrEdQueryExecutor(ctx, len(s), len(q), q, eps, alpha, beta)
            assert e.run() == len(off)
            assert np.array_equal(e.answers.offsets, off) and np.array_equal(e.answers.distances, dist)
            assert ctx.get_option(kvmatch.OPT_UCR_LOOPS) == 0               # the option does not outlive the run
            # the engine's own loop (`dist <= bsf`) on the same single interval: without the tie window
            L, R = np.array([1], dtype=np.int32), np.array([len(s) - len(q) + 1], dtype=np.int32)
            compare(kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, eps, alpha, beta), oracle.cnsm_ed(s, q, L, R, 0, eps, alpha, beta),
                    tag=f"engine on the tie instance ab={ab}")
        ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 1)
    finally:
        ctx.load_series(series)


def test_ucr_dtw_stale_cb_tie(world):
    """U/UcrDtwQueryExecutor.java never clears cb1 / cb2 between windows; at an exact LB_Keogh tie the DP then reads what
    earlier windows left there and rejects windows the engines (fresh arrays) report at distance eps
    (tests/test_oracle.py::test_ucr_dtw_executor_reads_stale_cb_entries_at_exact_ties).  Since round 3 the executor mirror
    reproduces that (KVM_OPT_UCR_LOOPS, kvm_ucr.hip): on this instance the executor's 8 windows, plain equality with the
    oracle's restatement -- while the ENGINE on the executor's segments keeps its 14 (the 6 extra ones all at distance eps)."""
    kvmatch, datagen, series, ctx = world
    from kvmatch import ucr
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ucr", "ucr_dtw_stale_cb_tie.npz"))
    s, q = g["s"].astype(np.float64), g["q"].astype(np.float64)
    eps, rho, alpha, beta = float(g["eps"]), int(g["rho"]), float(g["alpha"]), float(g["beta"])
    ctx.load_series(s)
    try:
        uo, ud, _ = oracle.ucr_dtw(s, q, eps, rho, alpha, beta)
        assert len(uo) == 8
        d = kvmatch.UcrDtwQueryExecutor(ctx, len(s), len(q), q, eps, rho, alpha, beta)
        assert d.run() == 8
        assert np.array_equal(d.answers.offsets, uo) and np.array_equal(d.answers.distances, ud)
        assert d.answers.stats[kvmatch.STAT_DTW_SEQUENTIAL] > 0 and ctx.get_option(kvmatch.OPT_UCR_LOOPS) == 0
        L, R = ucr.segments(1, len(s) - len(q) + 1, ucr.EPOCH - len(q) + 1)
        eng = kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, eps, rho, alpha, beta, resolved=True)
        assert np.array_equal(eng.offsets - 1, g["engine_off"]) and np.array_equal(eng.distances, g["engine_dist"]) and len(eng) == 14
        extra = set(zip((eng.offsets - 1).tolist(), eng.distances.tolist())) - set(zip(uo.tolist(), ud.tolist()))
        assert len(extra) == 6 and all(dist == eps for _, dist in extra)
    finally:
        ctx.load_series(series)


def test_residue_of_a_spike_in_the_references_running_sums(built):
    """The reference's running sums (ex += d, ex2 += d * d, later -= the same, K/NormQueryEngine.java:497-524) keep the
    rounding residue of everything that passed through them since the interval began: after a burst of 1e9-scale values --
    or O(1) values in a 1e-158-scaled series -- the mean and sigma of every later window of that interval are noise, and
    so are the answers; the oracle reproduces them, and so must the library (its pruners know the true sigma: the guard
    has to widen by what the run has seen, kvm_hist_*).  Found by the adversarial sweep, seed 23."""
    import kvmatch
    rng = np.random.default_rng(2323)
    n, m = 60_000, 256
    L, R = np.array([1], dtype=np.int32), np.array([n - m + 1], dtype=np.int32)       # one interval: one chain of sums
    with kvmatch.Phase2Context(0) as ctx:
        for scale, burst in ((1.0, 1e9), (1e-158, 1.0), (1.0, 1e5), (1e3, 1e12)):
            s = np.cumsum(rng.uniform(-1, 1, n)) * scale
            s[20_000:20_010] = burst * rng.uniform(0.5, 1.5, 10)
            q = s[40_000:40_000 + m].copy()
            ctx.load_series(s)
            for eps in (float(np.sqrt(m)), 8.0, 1.0):
                for ab in (1, 0):
                    ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
                    compare(kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, eps, 1e9, 1e9), oracle.cnsm_ed(s, q, L, R, 0, eps, 1e9, 1e9),
                            tag=f"spike ed scale={scale} burst={burst} eps={eps} ab={ab}")
                compare(kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, eps, 1.5, 3.0 * scale), oracle.cnsm_ed(s, q, L, R, 0, eps, 1.5, 3.0 * scale),
                        tag=f"spike ed filter scale={scale} burst={burst} eps={eps}")
            ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 1)
            Ld, Rd = np.array([19_000], dtype=np.int32), np.array([23_000], dtype=np.int32)   # (DTW on a shorter interval)
            for eps in (8.0, 2.0):
                compare(kvmatch.NormQueryEngineDtw(ctx).verify((Ld, Rd), 1, q, eps, 12, 1e9, 1e9), oracle.cnsm_dtw(s, q, Ld, Rd, 0, eps, 12, 1e9, 1e9),
                        tag=f"spike dtw scale={scale} burst={burst} eps={eps}")


def test_ed_pruner_sixteen_per_lane_form_on_every_list_shape(world):
    """Class A of the ED pruner on 16 candidates per lane (round 5: 16 rows x 4 lanes, ring-staged rows, compile-time loops) is
    chosen by the launcher for long lists only; here it is FORCED (KVM_OPT_ED_FORCE: the form of class A on host-planned lists;
    device-built cNSM lists of any size through it, never split) with early abandoning off, over window lengths that do and do
    not end on a chunk boundary (25 .. 4097: the last revolution runs on the zeros behind the query), every window split, lists
    of short, long and single-candidate runs, a shifted list and runs that end on the last window of the series.  Answers,
    order and every distance equal the oracle's; so does the 8-per-lane form's on the same calls.  What ran is read back
    (KVM_OPT_ED_LAST_SHAPE)."""
    kvmatch, datagen, series, ctx = world
    F = kvmatch.ed_force
    shape = lambda c=None: kvmatch.ed_shape((c or ctx).get_option(kvmatch.OPT_ED_LAST_SHAPE))  # noqa: E731
    ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0)
    try:
        n = len(series)
        for m, eps in ((25, 1.5), (100, 2.5), (128, 3.0), (129, 3.0), (300, 4.0), (1024, 9.0), (4097, 12.0)):
            s, q, L, R = planted(world, m)
            lists = [("mean run 48", L, R, 1)]
            l2, r2 = datagen.gen_intervals(n, m, 60_000, 400.0, must_cover=[planted.qoff, 40_000, 200_000, 250_000], seed=5)
            lists.append(("mean run 400", l2, r2, 1))
            l3 = np.unique(np.concatenate([np.arange(1, n - m, 97, dtype=np.int32), np.array([planted.qoff], np.int32)])).astype(np.int32)
            l3 = l3[np.concatenate(([True], np.diff(l3) > 1))]        # (disjoint, non-adjacent: the form phase 1 hands over)
            lists.append(("single candidates", l3, l3.copy(), 1))
            # one run that reaches the last window of the series, and the first
            lists.append(("ends of the series", np.array([1, 40_000 - 30, n - m + 1 - 200], np.int32), np.array([70, 40_000 + 40, n - m + 1], np.int32), 1))
            lists.append(("lastSegment 3", (L + 50).astype(np.int32), (R + 50).astype(np.int32), 3))
            for name, ll, rr, ls in lists:
                shift = (ls - 1) * 25
                want_r = oracle.rsm_ed(s, q, ll, rr, shift, eps)
                want_c = oracle.cnsm_ed(s, q, ll, rr, shift, eps, 1.5, 3.0)
                # ("16+8": class B on the four-lane rows as well, 8 candidates per lane)
                for form, a16, b16 in (("16+8", 2, 2), ("16", 2, 1), ("8", 1, 1)):
                    for split in ((1, 2, 4) if form != "8" and name != "single candidates" else (0,)):
                        ctx.set_option(kvmatch.OPT_ED_FORCE, F(split_a=split, split_b=split, a16=a16, b16=b16, dev_a16_any=1, dev_no_split=1))
                        compare(kvmatch.QueryEngine(ctx).verify((ll, rr), ls, q, eps), want_r, tag=f"rsm m={m} {name} form={form} split={split}")
                    ctx.set_option(kvmatch.OPT_ED_FORCE, F(a16=a16, b16=b16, dev_a16_any=1, dev_no_split=1))
                    compare(kvmatch.NormQueryEngine(ctx).verify((ll, rr), ls, q, eps, 1.5, 3.0), want_c, tag=f"cnsm m={m} {name} form={form}")
        # which form ran
        ctx.set_option(kvmatch.OPT_ED_FORCE, F(a16=2, b16=2))
        kvmatch.QueryEngine(ctx).verify((L, R), 1, q, eps)
        assert (shape()["per_lane_a"], shape()["per_lane_b"]) == (16, 8)
        ctx.set_option(kvmatch.OPT_ED_FORCE, F(a16=1))
        kvmatch.QueryEngine(ctx).verify((L, R), 1, q, eps)
        assert (shape()["per_lane_a"], shape()["per_lane_b"]) == (8, 4)
        # Left to itself the launcher takes class B to the four-lane rows where other queries run next to this one: told by
        # the caller (KVM_OPT_IN_FLIGHT) or by kvm_run_jobs for the contexts of its call -- and only for its duration
        ctx.set_option(kvmatch.OPT_ED_FORCE, F(a16=2))
        want = oracle.rsm_ed(s, q, L, R, 0, eps)
        kvmatch.QueryEngine(ctx).verify((L, R), 1, q, eps)
        assert shape()["per_lane_b"] == 4 and shape()["in_flight"] == 1
        ctx.set_option(kvmatch.OPT_IN_FLIGHT, 8)
        assert ctx.get_option(kvmatch.OPT_IN_FLIGHT) == 8
        compare(kvmatch.QueryEngine(ctx).verify((L, R), 1, q, eps), want, tag="KVM_OPT_IN_FLIGHT 8")
        assert shape()["per_lane_b"] == 8 and shape()["in_flight"] == 8
        ctx.set_option(kvmatch.OPT_IN_FLIGHT, 1)
        qs = kvmatch.QueryStream(ctx, in_flight=3)
        try:
            jobs = [kvmatch.Job(0, (L, R), 1, q, eps) for _ in range(6)]
            for got in qs.run_jobs(jobs):
                compare(got, want, tag="run_jobs, 3 in flight")
            for c in qs.ctxs:     # (the contexts of the call were told for its duration: every launch planned for 3 in flight)
                assert shape(c)["in_flight"] == 3 and shape(c)["per_lane_b"] == 8
            for got in qs.run([(lambda c: kvmatch.QueryEngine(c).verify((L, R), 1, q, eps)) for _ in range(2)]):
                compare(got, want, tag="QueryStream.run, 2 jobs on 3 contexts")
            assert sum(shape(c)["in_flight"] == 2 for c in qs.ctxs) == 2
        finally:
            qs.close()
        kvmatch.QueryEngine(ctx).verify((L, R), 1, q, eps)
        assert shape()["in_flight"] == 1
    finally:
        ctx.set_option(kvmatch.OPT_ED_FORCE, 0)
        ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 1)
        ctx.load_series(series)


def test_streams_with_the_enqueue_lock_forced(built):
    """KVM_OPT_SERIALIZE_ENQUEUE: the calls of a host that runs under a tool hooking the kernel dispatches take turns enqueueing
    (kvm_enqueue_lock, kvm_verify.hip: what lets `rocprofv3 --kernel-trace -- python3 bench.py` complete every time).  Its
    default comes from the environment once per process: a child process runs a mixed stream of all four variants through kvm_run_jobs and through the
    device-resident slots with KVM_SERIALIZE_ENQUEUE=1 -- every lock / release / re-lock path, the DTW calls' wait in the
    middle included -- and compares with the oracle."""
    import subprocess
    import sys
    env = dict(os.environ, KVM_SERIALIZE_ENQUEUE="1")
    cp = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "enqueue_lock_worker.py")], env=env,
                        capture_output=True, text=True, timeout=600)
    assert cp.returncode == 0 and "equal the oracle" in cp.stdout, cp.stdout[-2000:] + cp.stderr[-4000:]


def test_run_jobs_is_the_same_stream_in_native_threads(world):
    """kvm_run_jobs (QueryStream.run_jobs): calls as data, one native host thread per context.  Every job's answers equal
    the oracle's (and therefore the one-call-at-a-time engines'), in job order, all four variants mixed in one stream;
    results that are not collected are released; a failing job raises what its engine would have raised."""
    kvmatch, datagen, series, ctx = world
    m = 256
    s, q, L, R = planted(world, m)
    try:
        want = [oracle.rsm_ed(s, q, L, R, 0, 6.0), oracle.cnsm_ed(s, q, L, R, 0, 3.0, 1.5, 2.0),
                oracle.rsm_dtw(s, q, L, R, 0, 6.0, 12), oracle.cnsm_dtw(s, q, L, R, 0, 3.0, 12, 1.5, 2.0)]
        mk = [lambda: kvmatch.Job(0, (L, R), 1, q, 6.0), lambda: kvmatch.Job(1, (L, R), 1, q, 3.0, alpha=1.5, beta=2.0),
              lambda: kvmatch.Job(2, (L, R), 1, q, 6.0, rho=12), lambda: kvmatch.Job(3, (L, R), 1, q, 3.0, rho=12, alpha=1.5, beta=2.0)]
        qs = kvmatch.QueryStream(ctx, in_flight=3)
        try:
            jobs = [mk[i % 4]() for i in range(22)]
            got = qs.run_jobs(jobs)
            assert len(got) == 22
            for i, g in enumerate(got):
                compare(g, want[i % 4], tag=f"job {i}")
            some = qs.run_jobs(jobs, collect=[3, -1])
            assert [x is not None for x in some] == [i in (3, 21) for i in range(22)]
            compare(some[3], want[3], tag="collected 3")
            compare(some[21], want[21 % 4], tag="collected last")
            assert qs.run_jobs([]) == []
            # a list the cNSM engines cannot index (not sorted by left): IllegalArgumentException in Java
            bad = kvmatch.Job(1, (L[::-1].copy(), R[::-1].copy()), 1, q, 3.0, alpha=1.5, beta=2.0)
            with pytest.raises(kvmatch.KvMatchIllegalArgument):
                qs.run_jobs([mk[0](), bad, mk[0]()])
            compare(qs.run_jobs([mk[0]()])[0], want[0], tag="after a failed stream")
            # the text raised is the failing job's own, whichever context ran it -- not what the failure above left in one
            # of the contexts (kvm_run_jobs hands the first failing job's message to the first context)
            for pos in range(5):
                stream = [mk[0]() for _ in range(5)]
                stream[pos] = kvmatch.Job(1, (L, R), 1, q, 3.0, alpha=-1.0, beta=2.0)
                with pytest.raises(kvmatch.KvMatchIllegalArgument, match="alpha must be > 0"):
                    qs.run_jobs(stream)
        finally:
            qs.close()
    finally:
        ctx.load_series(series)

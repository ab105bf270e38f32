"""GPU parity at BASELINE.json's FULL sizes (-m gpu): n = 1e8 points, m = 8192 / 4096, 1e6 candidates in phase-1-shaped
intervals (configs[1..3]).  The oracle (with the reference's early abandoning) is fast enough on these lists to compare
against directly; on top of that, size-independent properties: the self match and every planted copy are found, the
answers do not depend on the pruning kernel, the early-abandon switch or on how the list is sharded, and cntCandidate is
additive over shards."""
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

N = 100_000_000


@pytest.fixture(scope="module")
def big(built):
    import kvmatch
    from kvmatch import datagen
    series = datagen.gen_series(N)
    m = 8192
    qoff = N // 3 + 17
    planted = [N // 7, N // 2 + 1000, N - 3 * m]
    q = datagen.plant_copies(series, qoff, m, planted, [1.0, 1.0, 1.0], [0.0, 0.02, -0.03], 0.01)
    ctx = kvmatch.Phase2Context(0)
    ctx.load_series(series)
    yield kvmatch, datagen, series, ctx, q, qoff, planted
    ctx.close()


def same(got, want):
    return len(got) == len(want[0]) and np.array_equal(got.offsets, want[0]) and np.array_equal(got.distances, want[1])


def test_rsm_ed_and_cnsm_ed_config_2_3(big):
    kvmatch, datagen, series, ctx, q, qoff, planted = big
    from kvmatch import distributed
    m = 8192
    L, R = datagen.gen_intervals(N, m, 1_000_000, 64.0, must_cover=[qoff] + planted, seed=1)
    assert int((R.astype(np.int64) - L + 1).sum()) >= 1_000_000
    want = oracle.rsm_ed(series, q, L, R, 0, 10.0)
    eng = kvmatch.QueryEngine(ctx)
    for kern in (0, 1, 3):             # every pruning kernel, abandon on and off: one answer list
        ctx.set_option(kvmatch.OPT_ED_KERNEL, kern)
        for ab in (1, 0):
            ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
            got = eng.verify((L, R), 1, q, 10.0)
            assert same(got, want), (kern, ab)
            assert got.cnt_candidate == want[2]["cnt_candidate"]
    ctx.set_option(kvmatch.OPT_ED_KERNEL, 0)
    ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 1)
    offs = set(want[0].tolist())
    assert qoff in offs and all(p in offs for p in planted)
    assert want[1][0] == 0.0 and want[0][0] == qoff                       # the self match, distance exactly 0
    # sharding invariance: the union of the shards' answers is the unsharded list, cntCandidate is additive
    for world in (2, 8):
        o, d, c = [], [], 0
        for rank in range(world):
            l, r = distributed.split_intervals(L, R, world, rank)
            a = eng.verify((l, r), 1, q, 10.0)
            o.append(a.offsets); d.append(a.distances); c += a.cnt_candidate
        so, sd = distributed.sort_answers(np.concatenate(o), np.concatenate(d))
        assert c == want[2]["cnt_candidate"] and np.array_equal(so, want[0]) and np.array_equal(sd, want[1])
    # the same list planned for throughput (round 5: queries in flight -> no window split beyond the launch's share of the chip,
    # class B on the four-lane rows): a stream of 8 through kvm_run_jobs and a single call under KVM_OPT_IN_FLIGHT, every point
    # evaluated (the forms exist only there) -- one answer list
    ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 0)
    qs = kvmatch.QueryStream(ctx, in_flight=8)
    try:
        for got in qs.run_jobs([kvmatch.Job(0, (L, R), 1, q, 10.0) for _ in range(16)]):
            assert same(got, want) and got.cnt_candidate == want[2]["cnt_candidate"]
    finally:
        qs.close()
    ctx.set_option(kvmatch.OPT_IN_FLIGHT, 8)
    assert same(eng.verify((L, R), 1, q, 10.0), want)
    ctx.set_option(kvmatch.OPT_IN_FLIGHT, 1)
    ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 1)
    # cNSM-ED, configs[2]: alpha / beta from SURVEY 8d
    neng = kvmatch.NormQueryEngine(ctx)
    for eps, alpha, beta in ((1.0, 1.1, 44.974 * 0.5), (10.0, 2.0, 10.0)):
        want = oracle.cnsm_ed(series, q, L, R, 0, eps, alpha, beta)
        for ab in (1, 0):
            ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
            got = neng.verify((L, R), 1, q, eps, alpha, beta)
            assert same(got, want), (eps, alpha, beta, ab)
            assert got.stats[1] == want[2]["pass_filter"]
        assert qoff in set(want[0].tolist())
    ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 1)


def test_dtw_config_4(big):
    kvmatch, datagen, series, ctx, q8, qoff, planted = big
    m, rho = 4096, 204
    q = series[qoff - 1: qoff - 1 + m].copy()
    L, R = datagen.gen_intervals(N, m, 100_000, 64.0, must_cover=[qoff, planted[1]], seed=2)
    want = oracle.cnsm_dtw(series, q, L, R, 0, 3.0, rho, 2.0, 10.0)
    for dp_form in (0, 3, 2):                                              # (3: two candidates per wave, 2: one; 0 picks by list length)
        ctx.set_option(kvmatch.OPT_DP_KERNEL, dp_form)
        got = kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, 3.0, 0.05, 2.0, 10.0)
        assert same(got, want) and got.cnt_candidate == want[2]["cnt_candidate"], dp_form
    ctx.set_option(kvmatch.OPT_DP_KERNEL, 0)
    assert qoff in set(want[0].tolist()) and want[1][0] == 0.0
    ctx.set_option(kvmatch.OPT_LB_CASCADE, 0)                              # the cascade is only an optimisation
    got = kvmatch.NormQueryEngineDtw(ctx).verify((L[:200], R[:200]), 1, q, 3.0, 0.05, 2.0, 10.0)
    ctx.set_option(kvmatch.OPT_LB_CASCADE, 1)
    assert same(got, oracle.cnsm_dtw(series, q, L[:200], R[:200], 0, 3.0, rho, 2.0, 10.0))
    want = oracle.rsm_dtw(series, q, L, R, 0, 6.0, rho)
    got = kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, 6.0, 0.05)
    assert same(got, want) and qoff in set(want[0].tolist())
    # The query above is a smooth slice: JDK 8's TimSort leaves it in index order.  A query on which the reference's sort
    # comes through with a NON-identity permutation exercises order[] -- the summation order of both LB_Keogh loops
    # (K/NormQueryEngineDtw.java:475-490) -- at BASELINE size: the first such slice of the series, found by asking the library's
    # restatement of the sort (which the oracle's must agree with, or the two sides below would not match).
    q2off = None
    for cand in range(1000, N - m, 7919):
        try:
            order = kvmatch.cnsm_dtw_order(series[cand - 1: cand - 1 + m])
        except kvmatch.KvMatchIllegalArgument:
            continue                                   # (the reference cannot run this one at all)
        moved = int((order != np.arange(m)).sum())
        if 0 < moved < m:                              # (not a plain reversal either)
            q2off = cand
            break
    assert q2off is not None
    q2 = series[q2off - 1: q2off - 1 + m].copy()
    L2, R2 = datagen.gen_intervals(N, m, 100_000, 64.0, must_cover=[q2off], seed=5)
    for eps in (3.0, 12.0):
        want = oracle.cnsm_dtw(series, q2, L2, R2, 0, eps, rho, 2.0, 10.0)
        got = kvmatch.NormQueryEngineDtw(ctx).verify((L2, R2), 1, q2, eps, 0.05, 2.0, 10.0)
        assert same(got, want) and got.cnt_candidate == want[2]["cnt_candidate"] and q2off in set(want[0].tolist()), (eps, moved)


@pytest.mark.parametrize("w", [25, 50, 400])
def test_index_build_n_1e8_equals_the_oracle(big, w):
    """The whole index build for one window length over the n = 1e8 series: the file image equals the oracle's byte for
    byte (3-6e7 step-1 intervals in ~5e4 rows, merged by step 2, ~1e8 bytes), and its statistic block accounts for every
    window position."""
    import struct
    import time
    kvmatch, datagen, series, ctx, q, qoff, planted = big
    t0 = time.perf_counter()
    got, info = ctx.index_build(w)
    t_lib = time.perf_counter() - t0
    t0 = time.perf_counter()
    want, winfo = oracle.index_file(series, w)
    t_cpu = (time.perf_counter() - t0) / 2          # (oracle.index_file runs the C function twice: size, then fill)
    assert len(got) == len(want) and got == want
    assert [info[k] for k in ("rows_before", "rows_after", "intervals_before", "intervals_after")] == winfo
    last_line = struct.unpack(">i", got[-4:])[0]
    offsets = struct.unpack(">%di" % ((len(got) - last_line) // 4), got[last_line:])
    assert len(offsets) == info["rows_after"] + 2
    key, cum_iv, cum_off = struct.unpack(">dii", got[offsets[-1] - 16: offsets[-1]])
    assert cum_iv == info["intervals_after"] and cum_off == N - w + 1
    print(f"index build n=1e8 w={w}: library {t_lib:.2f} s (step 1 {info['step1_ms']:.0f} ms, step 2 {info['step2_ms']:.0f} ms, "
          f"encode {info['encode_ms']:.0f} ms), oracle on one core {t_cpu:.1f} s, {len(got)} bytes, "
          f"{info['rows_before']} -> {info['rows_after']} rows")


def test_phase1_should_stop_is_a_pure_function(big):
    """kvm_phase1_should_stop -- the reference's early-termination test (K/QueryEngine.java:316-327) on the library's time
    model -- is a pure function of its arguments (the reference's is not: wall clock).  The model's accuracy against
    measured calls is a wall-clock matter and lives in tests/test_zz_perf_gpu.py, behind every parity file."""
    kvmatch = big[0]
    last, stops = 0.0, []
    for i in range(12):
        stop, last = kvmatch.phase1_should_stop(0, i, 0.02 * (i + 1), 4000 >> min(i, 10), 4_000_000 >> i if i < 6 else 62_500, 8192, last)
        stops.append(stop)
        if stop:
            break
    again, last2 = [], 0.0
    for i in range(len(stops)):
        stop, last2 = kvmatch.phase1_should_stop(0, i, 0.02 * (i + 1), 4000 >> min(i, 10), 4_000_000 >> i if i < 6 else 62_500, 8192, last2)
        again.append(stop)
    assert stops == again and stops[-1] and not any(stops[:5])


# BETA_BASE of K/experiments/NormQueryDtwSelectivityGenerate.java:38, indexed by log2(m) - 8
BETA_BASE = (12.663, 19.651, 24.289, 44.974, 56.526)


@pytest.mark.parametrize("variant", ["cnsm_ed", "cnsm_dtw"])
def test_the_references_own_parameter_grid_at_full_size(big, variant):
    """The grid the reference's experiment driver sweeps (K/experiments/NormQueryDtwSelectivityGenerate.java:38,63-93):
    alpha in {1.1, 1.5, 2.0} x beta = BETA_BASE[log2(m) - 8] * {0.5, 1, 5, 10} x eps in {1, 10}, ten seeded query offsets per
    setting -- cNSM-ED and cNSM-DTW at m = 4096 (rho = floor(0.05 m) = 204; the table has five entries, so m = 4096 is its
    last) over the n = 1e8 series, the reference's early abandoning on.  Every query: answers (offsets, order, every bit of
    every distance), cntCandidate and the stage counters equal the oracle's.  The oracle runs on the host's cores in
    parallel (ctypes releases the interpreter lock); the list is 1e5 candidates per query so that 240 oracle runs fit a test."""
    import concurrent.futures as cf
    kvmatch, datagen, series, ctx, q8, qoff8, planted = big
    m, rho = 4096, 204
    beta_base = BETA_BASE[int(np.log2(m)) - 8]
    rng = np.random.default_rng(20241004)
    settings = [(eps, alpha, beta_base * f) for eps in (1.0, 10.0) for alpha in (1.1, 1.5, 2.0) for f in (0.5, 1.0, 5.0, 10.0)]
    jobs = []
    for si, (eps, alpha, beta) in enumerate(settings):
        for t in range(10):
            qo = int(rng.integers(1, N - m))
            if variant == "cnsm_dtw":   # (the reference can only run queries its Arrays.sort accepts: conftest.sortable_query)
                from conftest import sortable_query
                qo = sortable_query(series, m, qo, step=100_003)   # (hop over whole noise segments, where that sort throws throughout)
            jobs.append((si, t, qo, eps, alpha, beta))
    lists = [datagen.gen_intervals(N, m, 100_000, 64.0, must_cover=[j[2] for j in jobs[k::4]], seed=40 + k) for k in range(4)]

    def cpu(job):
        si, t, qo, eps, alpha, beta = job
        L, R = lists[(si * 10 + t) % 4]
        q = series[qo - 1: qo - 1 + m]
        return (oracle.cnsm_ed(series, q, L, R, 0, eps, alpha, beta) if variant == "cnsm_ed"
                else oracle.cnsm_dtw(series, q, L, R, 0, eps, rho, alpha, beta))

    with cf.ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as pool:
        wants = list(pool.map(cpu, jobs))
    eng = kvmatch.NormQueryEngine(ctx) if variant == "cnsm_ed" else kvmatch.NormQueryEngineDtw(ctx)
    n_answers = 0
    for job, want in zip(jobs, wants):
        si, t, qo, eps, alpha, beta = job
        L, R = lists[(si * 10 + t) % 4]
        q = series[qo - 1: qo - 1 + m].copy()
        got = eng.verify((L, R), 1, q, eps, alpha, beta) if variant == "cnsm_ed" else eng.verify((L, R), 1, q, eps, rho, alpha, beta)
        st = want[2]
        assert same(got, want), job
        assert got.cnt_candidate == st["cnt_candidate"] and got.stats[0] == st["windows"] and got.stats[1] == st["pass_filter"], job
        if variant == "cnsm_dtw":
            # the screens in front of the exact stage may keep what the reference drops (never the reverse); the exact stage
            # decides with the reference's own values, so the DPs it starts are the reference's dtw() calls
            assert got.stats[2] >= st["pass_kim"] and got.stats[3] >= st["pass_keogh"], (job, got.stats, st)
            assert got.stats[4] == st["pass_keogh2"], (job, got.stats, st)
        assert qo in set(want[0].tolist()), job      # the self match (K/experiments/NormQueryTestGroupBySelectivity.java:101-103)
        n_answers += len(got)
    assert n_answers >= len(jobs)

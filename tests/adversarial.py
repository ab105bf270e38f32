"""Adversarial parity cases shared by tests/test_adversarial_gpu.py and profiles/stress_adversarial.py: inputs chosen
to sit ON the decision boundaries or to break naive arithmetic -- exact ties with eps^2 (integer-valued data),
alpha = 1 / beta = 0 filters around exact copies, piecewise-constant data (sigma = 0), huge level offsets (cancellation),
wide dynamic range (a burst leaves the prefix sums of its block large), tiny and odd query lengths, single-offset
intervals, eps = 0, NaN / Inf points in the series -- all four variants against the oracle, bit for bit.

The reference's DTW tie semantics are part of the bar: DtwUtils.dtw returns `min_cost + cb[i+r+1]` as soon as that
lower bound is `>= bsf` (K/utils/DtwUtils.java:324-326), the LB loops stop at the first partial sum `>= bsf` and the
callers accept `<= eps^2`, so a bound that EQUALS eps^2 exactly makes the candidate an answer at distance eps although its
DTW distance is larger.  On continuous data that tie has measure zero; on integer / half-integer data with an integer
eps^2 it is common.  The library must return exactly the oracle's list there too (`ties` counts the verifications in
which it had to take its sequential reference-ordered path, KVM_STAT_DTW_SEQUENTIAL)."""
import time

import numpy as np

KINDS = ("ints", "steps", "offset", "range", "halfint", "nan", "tiny", "huge")


def make(kind, n, r):
    if kind == "ints":      # small integers: every distance is an exactly representable integer
        return r.integers(-6, 7, n).astype(np.float64)
    if kind == "steps":     # piecewise constant: long windows with sigma == 0, many exact ties
        return np.repeat(r.integers(-3, 4, n // 97 + 1).astype(np.float64), 97)[:n]
    if kind == "offset":    # random walk riding on a huge level
        return np.cumsum(r.uniform(-1, 1, n)) + 1e7
    if kind == "range":     # wide dynamic range
        return r.normal(0, 1, n) * np.exp(r.uniform(-6, 8, n // 1000 + 1).repeat(1000)[:n])
    if kind == "halfint":   # multiples of 0.5 around zero: exact products and sums
        return r.integers(-9, 10, n) * 0.5
    if kind == "nan":       # a handful of NaN / Inf points in an otherwise ordinary series
        s = np.cumsum(r.uniform(-1, 1, n)) * 0.1
        idx = r.integers(0, n, 6)
        s[idx[:3]] = np.nan; s[idx[3:5]] = np.inf; s[idx[5]] = -np.inf
        return s
    if kind == "tiny":      # squares underflow to zero / denormals
        return np.cumsum(r.uniform(-1, 1, n)) * 1e-158
    if kind == "huge":      # squares overflow to infinity in places
        return np.cumsum(r.uniform(-1, 1, n)) * np.where(r.uniform(0, 1, n // 500 + 1).repeat(500)[:n] < 0.2, 1e153, 1e3)
    raise ValueError(kind)


def cases(datagen, rounds, seed, kinds=KINDS, n_range=(30_000, 120_000)):
    """The seeded sequence of adversarial cases: yields (rd, kind, n, m, rho, s, q, qoff, pl, lists, eps_tie, eps_generic)
    with lists = [(name, L, R), ...] and eps_generic one value per list."""
    rng = np.random.default_rng(seed)
    for rd in range(rounds):
        for kind in kinds:
            n = int(rng.integers(*n_range))
            s = make(kind, n, rng)
            m = int(rng.choice([25, 31, 64, 100, 257, 1024, 2500]))
            rho = int(rng.choice([0, 1, max(1, m // 20), max(1, m // 10)]))
            for attempt in range(50):       # a query window that is finite and not constant (stdQ > 0)
                qoff = int(rng.integers(1, n - m + 1))
                w = s[qoff - 1: qoff - 1 + m]
                if np.all(np.isfinite(w)) and np.std(w) > 0:
                    break
            pl = [int(p) for p in rng.integers(1, n - m + 1, 3)]     # exact copies (noise 0), the third shifted by +1
            q = datagen.plant_copies(s, qoff, m, pl, [1.0, 1.0, 1.0], [0.0, 0.0, 1.0], 0.0, seed=rd)
            lists = []
            L, R = datagen.gen_intervals(n, m, int(rng.integers(500, 8000)), float(rng.choice([1.0, 8.0, 64.0, 400.0])),
                                         must_cover=[qoff] + pl, seed=rd)
            lists.append(("gen", L, R))
            pts = np.unique(np.concatenate([rng.integers(1, n - m + 1, 300), [qoff] + pl])).astype(np.int32)
            pts = pts[np.concatenate(([True], np.diff(pts) > 1))]       # single-offset intervals, non-adjacent
            lists.append(("points", pts, pts.copy()))
            lists.append(("full", np.array([1], dtype=np.int32), np.array([n - m + 1], dtype=np.int32)))
            # thresholds: zero, an exactly achievable distance (the +1 shifted copy: eps^2 == m on exact data), a generic one
            d0 = float(np.sum((s[pl[2] - 1: pl[2] - 1 + m] - q) ** 2))
            eps_tie = float(np.sqrt(d0)) if np.isfinite(d0) and d0 > 0 and np.sqrt(d0) ** 2 == d0 else 1.0
            eps_generic = [float(rng.uniform(0.5, 4.0)) for _ in lists]
            yield rd, kind, n, m, rho, s, q, qoff, pl, lists, eps_tie, eps_generic


def run(kvmatch, datagen, oracle, ctx, rounds, seed, kinds=KINDS, n_range=(30_000, 120_000), log=None):
    """Returns (verifications, mismatches, DTW verifications that needed the sequential tie path)."""
    total = bad = quirk = thrown = 0
    ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 0)
    oracle.set_lenient_sort(False)
    t_gpu = t_cpu = 0.0

    def gpu(fn):
        nonlocal t_gpu
        t = time.perf_counter()
        try:
            r = fn()
        except kvmatch.KvMatchIllegalArgument as e:       # (cNSM-DTW: the reference's Arrays.sort throws on this query)
            r = ("illegal", str(e))
        dt = time.perf_counter() - t; t_gpu += dt
        if log and dt > 0.25 and not isinstance(r, tuple):
            log(f"  slow library call: {dt:.2f} s, {len(r)} answers of {r.cnt_candidate} candidates, stats {r.stats}")
        return r

    def cpu(fn):
        nonlocal t_cpu
        t = time.perf_counter()
        try:
            r = fn()
        except oracle.OracleError as e:
            if e.code != -2:
                raise
            r = ("illegal", str(e))
        t_cpu += time.perf_counter() - t
        return r

    def same(a_fn, o_fn, tag, eps, dtw):
        nonlocal total, bad, quirk, thrown
        total += 1
        a, o = gpu(a_fn), cpu(o_fn)
        if isinstance(a, tuple) or isinstance(o[0], str):       # an IllegalArgumentException: on both sides or it is a mismatch
            if not (isinstance(a, tuple) and isinstance(o[0], str)):
                bad += 1
                if log:
                    log(f"MISMATCH {tag}: only one side threw ({a if isinstance(a, tuple) else o})")
                return
            # the reference's sort threw on this query, on both sides: compare the arithmetic with the switch that carries on
            thrown += 1
            ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 1)
            prev = oracle.set_lenient_sort(True)
            try:
                a, o = gpu(a_fn), cpu(o_fn)
            finally:
                ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 0)
                oracle.set_lenient_sort(prev)
        oo, od, oc = o[0], o[1], o[2]
        ok = len(a) == len(oo) and np.array_equal(a.offsets, oo) and np.array_equal(a.distances, od, equal_nan=True) \
            and a.cnt_candidate == oc["cnt_candidate"]
        if dtw and a.stats[7] > 0:
            quirk += 1
        if ok:
            return
        bad += 1
        if log:
            log(f"MISMATCH {tag} got {len(a)} want {len(oo)}")

    last_rd = -1
    for rd, kind, n, m, rho, s, q, qoff, pl, lists, eps_tie, eps_generic in cases(datagen, rounds, seed, kinds, n_range):
        if rd != last_rd and last_rd >= 0 and log:
            log(f"round {last_rd}: verifications {total} mismatches {bad} DTW verifications through the sequential tie path {quirk}")
        last_rd = rd
        if True:
            ctx.load_series(s)
            for li, (name, L, R) in enumerate(lists):
                for eps in (0.0, eps_tie, eps_generic[li]):
                    for alpha, beta in ((1.0, 0.0), (1.0, 1.0), (1.5, 0.5), (1e9, 1e9)):
                        tag = f"{kind} rd={rd} n={n} m={m} list={name} eps={eps:.4f} a={alpha} b={beta} rho={rho}"
                        for ab in (1, 0):
                            ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
                            if alpha == 1.0 and beta == 0.0:
                                same((lambda: kvmatch.QueryEngine(ctx).verify((L, R), 1, q, eps)), (lambda: oracle.rsm_ed(s, q, L, R, 0, eps)),
                                     "rsm_ed " + tag + f" ab={ab}", eps, False)
                            same((lambda: kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, eps, alpha, beta)),
                                 (lambda: oracle.cnsm_ed(s, q, L, R, 0, eps, alpha, beta)), "cnsm_ed " + tag + f" ab={ab}", eps, False)
                        if name != "full" or n < 60_000:
                            ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 1)
                            if alpha == 1.0 and beta == 0.0:
                                r2 = int(rho)        # (an int rho is the band half-width in points)
                                same((lambda: kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, eps, r2)),
                                     (lambda: oracle.rsm_dtw(s, q, L, R, 0, eps, r2)), "rsm_dtw " + tag, eps, True)
                            same((lambda: kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, eps, rho, alpha, beta, resolved=True)),
                                 (lambda: oracle.cnsm_dtw(s, q, L, R, 0, eps, rho, alpha, beta)), "cnsm_dtw " + tag, eps, True)
            if log:
                log(f"  {kind} rd={rd} n={n} m={m}: library {t_gpu:.1f} s, oracle {t_cpu:.1f} s so far")
    if log and last_rd >= 0:
        log(f"round {last_rd}: verifications {total} mismatches {bad} DTW verifications through the sequential tie path {quirk}; "
            f"queries on which the reference's sort throws (both sides, then compared with the switch on) {thrown}")
    ctx.set_option(kvmatch.OPT_EARLY_ABANDON, 1)
    return total, bad, quirk

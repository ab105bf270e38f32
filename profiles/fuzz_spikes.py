#!/usr/bin/env python3
"""cNSM engines on series with bursts of every size (1e2 .. 1e10 x the local sigma, 1 .. 2m points long) inside long
intervals: the reference's running sums carry the residue of the burst through the rest of the interval, and the pruners'
guard has to cover exactly that (kvm_hist_*) -- library vs oracle.  `fuzz_spikes.py [cases] [seed]`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, kvmatch, oracle
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
oracle.set_lenient_sort(True)
bad = total = 0
with kvmatch.Phase2Context(0) as ctx:
    ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 1)
    for case in range(cases):
        n = int(rng.choice([20_000, 50_000])); m = int(rng.choice([64, 256, 1024]))
        level = float(rng.choice([0.0, 10.0, 1e4]))
        s = level + np.cumsum(rng.uniform(-1, 1, n)) * float(rng.choice([1.0, 1e-3, 1e3]))
        sig = float(np.std(np.diff(s))) * np.sqrt(m)
        for _ in range(int(rng.integers(1, 4))):
            p = int(rng.integers(0, n - 2 * m)); ln = int(rng.choice([1, 3, m // 2, m, 2 * m]))
            s[p:p + ln] += sig * float(10.0 ** rng.uniform(2, 10)) * rng.uniform(0.5, 1.0, ln) * rng.choice([-1.0, 1.0])
        qoff = int(rng.integers(1, n - m)); q = s[qoff - 1:qoff - 1 + m].copy()
        # plant near-copies AFTER bursts so that there are answers whose sigma the reference has polluted
        for p in rng.integers(1, n - m, 4):
            s[p - 1:p - 1 + m] = q * rng.uniform(0.9, 1.1) + rng.normal(size=m) * 0.02 * np.std(q)
        ctx.load_series(s)
        L, R = np.array([1], dtype=np.int32), np.array([n - m + 1], dtype=np.int32)
        for eps in (float(rng.choice([0.5, 2.0])), float(np.sqrt(m)) * float(rng.choice([0.3, 1.0]))):
            a, b = float(rng.choice([1.2, 1e9])), float(rng.choice([np.std(q), 1e9 * (1 + abs(level))]))
            total += 1
            g = kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, eps, a, b); o = oracle.cnsm_ed(s, q, L, R, 0, eps, a, b)
            if not (np.array_equal(g.offsets, o[0]) and np.array_equal(g.distances, o[1])):
                bad += 1; print(f"DIFFER cnsm_ed case {case} n={n} m={m} level={level} eps={eps:.3f} a={a} b={b:.3g}: {len(g)} vs {len(o[0])}", flush=True)
        Ld, Rd = np.array([max(1, qoff - 1500)], dtype=np.int32), np.array([min(n - m + 1, qoff + 1500)], dtype=np.int32)
        eps = float(rng.choice([1.0, 4.0])); total += 1
        g = kvmatch.NormQueryEngineDtw(ctx).verify((Ld, Rd), 1, q, eps, m // 20, 1e9, 1e9 * (1 + abs(level)), resolved=True)
        o = oracle.cnsm_dtw(s, q, Ld, Rd, 0, eps, m // 20, 1e9, 1e9 * (1 + abs(level)))
        if not (np.array_equal(g.offsets, o[0]) and np.array_equal(g.distances, o[1])):
            bad += 1; print(f"DIFFER cnsm_dtw case {case} n={n} m={m} level={level} eps={eps}: {len(g)} vs {len(o[0])}", flush=True)
print("engine calls", total, "differences", bad)
sys.exit(1 if bad else 0)

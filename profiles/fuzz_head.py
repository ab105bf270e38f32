#!/usr/bin/env python3
"""RSM-ED with early abandoning on long windows (m >= 2048), where what the pruner's early tests leave alive decides who
evaluates it (the HAND-OVER rule of kvm_ed_dot.hip: a wave left with a handful of live candidates gives them to the exact
evaluation): random and hostile series, lists of phase-1 shape and single offsets, epsilon ON candidates' distances (taken
from the oracle's wide-open list) and one ulp to either side, plus thresholds below which everything dies at once and
above which nothing does.  Library vs oracle.  `fuzz_head.py [cases] [seed]`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, kvmatch, oracle
from kvmatch import datagen
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
bad = total = 0
with kvmatch.Phase2Context(0) as ctx:
    for case in range(cases):
        n = int(rng.choice([30_000, 80_000]))
        kind = case % 5
        if kind == 0: s = datagen.gen_series(n, seed=seed * 1000 + case)
        elif kind == 1: s = rng.integers(-3, 4, n).astype(np.float64)
        elif kind == 2: s = np.repeat(rng.normal(size=n // 50 + 1), 50)[:n].copy()
        elif kind == 3: s = np.tile(rng.normal(size=128), n // 128 + 1)[:n] + (rng.random(n) < 0.002) * rng.normal(size=n) * 3
        else: s = 1e6 + np.cumsum(rng.normal(size=n))
        m = int(rng.choice([2048, 2049, 2304, 3000, 4096, 5000]))
        qoff = int(rng.integers(1, n - m))
        q = s[qoff - 1:qoff - 1 + m].copy()
        if rng.random() < 0.5: q = q + rng.normal(size=m) * 0.01
        ctx.load_series(s)
        if rng.random() < 0.5:
            L, R = datagen.gen_intervals(n, m, int(rng.integers(200, (n - m) // 3)), float(rng.choice([1.0, 8.0, 40.0])), must_cover=[qoff], seed=case)
        else:
            L = np.sort(rng.integers(1, n - m, int(rng.integers(1, 300)))).astype(np.int64); R = L + rng.integers(0, 3, len(L))
        wide = oracle.rsm_ed(s, q, L, R, 0, 1e300)
        d = np.sort(wide[1])
        picks = [0.0, float(d[0]) if len(d) else 1.0, float(d[len(d) // 2]) if len(d) else 2.0, float(d[-1]) if len(d) else 3.0,
                 float(np.quantile(d, 0.01)) if len(d) else 4.0]
        eps_list = []
        for x in picks:
            eps_list += [x, float(np.nextafter(x, 0.0)), float(np.nextafter(x, np.inf))]
        for eps in eps_list:
            if not (eps >= 0 and np.isfinite(eps)): continue
            want = oracle.rsm_ed(s, q, L, R, 0, eps)
            got = kvmatch.QueryEngine(ctx).verify((L, R), 1, q, eps)
            total += 1
            ok = (np.array_equal(got.offsets, want[0]) and np.array_equal(got.distances, want[1])
                  and got.cnt_candidate == want[2]["cnt_candidate"])
            if not ok:
                bad += 1
                print(f"DIFFER case {case} kind {kind} n={n} m={m} k={len(L)} eps={eps!r}: {len(got)} vs {len(want[0])}", flush=True)
        if case % 10 == 9: print(f"case {case + 1}: {total} queries, {bad} differences", flush=True)
print(f"fuzz_head: {total} queries, {bad} differences")
sys.exit(1 if bad else 0)

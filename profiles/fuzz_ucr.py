#!/usr/bin/env python3
"""The index-free full-scan executors (kvmatch.UcrEdQueryExecutor / UcrDtwQueryExecutor) against the oracle's
restatement of U/UcrEdQueryExecutor.java / U/UcrDtwQueryExecutor.java on random small series: n around the EPOCH
buffer boundaries, integer-valued data (ties), planted copies.  `fuzz_ucr.py [cases] [seed] [mix|int]` (int: every series
integer-valued -- where partial sums land exactly on eps^2: the executors' own loop artefacts, reproduced since round 3 by
KVM_OPT_UCR_LOOPS; plain equality is asserted, nothing is counted apart any more)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, kvmatch, oracle
from kvmatch import datagen
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
only_int = len(sys.argv) > 3 and sys.argv[3] == "int"
oracle.set_lenient_sort(True)
bad = total = ties = seq = 0
with kvmatch.Phase2Context(0) as ctx:
    ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 1)
    for case in range(cases):
        n = int(rng.choice([3_000, 20_000, 100_000, 100_001, 120_000] if only_int else [3_000, 20_000, 100_000, 100_001, 199_000, 230_000]))
        m = int(rng.choice([25, 60, 128, 300]))
        kind = 2 if only_int else case % 3
        s = datagen.gen_series(n, seed=case + 5) if kind == 0 else (np.cumsum(rng.normal(size=n)) * 0.1 if kind == 1 else rng.integers(-3, 4, n).astype(np.float64))
        qoff = int(rng.integers(1, n - m))
        q = s[qoff - 1:qoff - 1 + m].copy()
        if q.std() == 0: q[0] += 1
        for p in rng.integers(1, n - m, 3):
            s[p - 1:p - 1 + m] = q * (1.0 if kind == 2 else rng.uniform(0.8, 1.2)) + (0.0 if kind == 2 else rng.uniform(-1, 1))
        ctx.load_series(s)
        eps = float(rng.choice([0.0, 0.5, 2.0])) * (np.sqrt(m) * 0.2 if kind != 2 else 1.0); rho = int(rng.choice([0, 2, m // 10]))
        a, b = float(rng.choice([1.0, 1.3, 1e9])), float(rng.choice([0.0, 1.5, 1e9]))
        total += 2
        e = kvmatch.UcrEdQueryExecutor(ctx, n, m, q, eps, a, b); e.run()
        w = oracle.ucr_ed(s, q, eps, a, b)
        ties += int(np.any((w[1] == eps) & (eps > 0)))   # (scans with an answer at distance exactly eps: tie-prone instances)
        if not (np.array_equal(e.answers.offsets, w[0]) and np.array_equal(e.answers.distances, w[1])):
            bad += 1; print(f"DIFFER ucr_ed case {case} n={n} m={m} eps={eps} a={a} b={b}: {len(e.answers)} vs {len(w[0])}", flush=True)
            if os.environ.get("FUZZ_DUMP"): np.savez(os.path.join(ROOT, "gpurun_out", f"ucr_case_{sys.argv[2] if len(sys.argv) > 2 else 1}_{case}_ed.npz"), s=s, q=q, eps=eps, a=a, b=b, lib_off=e.answers.offsets, lib_dist=e.answers.distances, orc_off=w[0], orc_dist=w[1])
        d = kvmatch.UcrDtwQueryExecutor(ctx, n, m, q, eps, rho, a, b); d.run()
        w = oracle.ucr_dtw(s, q, eps, rho, a, b)
        seq += int(d.answers.stats[kvmatch.STAT_DTW_SEQUENTIAL] > 0)
        if case % 50 == 49: print(f"... {case + 1} cases, differences so far {bad}", flush=True)
        if not (np.array_equal(d.answers.offsets, w[0]) and np.array_equal(d.answers.distances, w[1])):
            bad += 1; print(f"DIFFER ucr_dtw case {case} n={n} m={m} eps={eps} rho={rho} a={a} b={b}: {len(d.answers)} vs {len(w[0])}", flush=True)
            if os.environ.get("FUZZ_DUMP"): np.savez(os.path.join(ROOT, "gpurun_out", f"ucr_case_{sys.argv[2] if len(sys.argv) > 2 else 1}_{case}_dtw.npz"), s=s, q=q, eps=eps, rho=rho, a=a, b=b, lib_off=d.answers.offsets, lib_dist=d.answers.distances, orc_off=w[0], orc_dist=w[1])
print("executor runs", total, "differences", bad, "| ED scans with an answer at distance exactly eps", ties, "| DTW scans through the sequential tie path", seq)
sys.exit(1 if bad else 0)

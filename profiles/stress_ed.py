#!/usr/bin/env python3
"""One-off stress run (not part of the test suite): random ED verifications with LONG queries (many chunks per unit,
window split, periodic early-abandon checks, cNSM retile) against the oracle, bit for bit.  `stress_ed.py [cases]`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, ROOT)
import numpy as np, kvmatch
from kvmatch import datagen
from oracle import oracle
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(os.environ.get("SEED", "7")))
base = datagen.gen_series(3_000_000)
ctx = kvmatch.Phase2Context(0)
bad = 0
def same(a, o, tag):
    global bad
    oo, od, oc = o[0], o[1], o[2]
    ok = len(a) == len(oo) and np.array_equal(a.offsets, oo) and np.array_equal(a.distances, od) and a.cnt_candidate == oc["cnt_candidate"]
    if not ok:
        bad += 1
        print("MISMATCH", tag, len(a), len(oo))
for c in range(cases):
    n = int(rng.integers(60_000, 400_000))
    kind = c % 3
    if kind == 0: s = base[int(rng.integers(0, 2_000_000)):][:n].copy()
    elif kind == 1: s = np.cumsum(rng.uniform(-1, 1, n)) + rng.uniform(-3000, 3000)
    else: s = 3 * np.sin(np.arange(n) * rng.uniform(0.0005, 0.05)) + rng.normal(0, 0.2, n)
    m = int(rng.choice([1300, 2048, 3000, 5000, 8192, 12000]))
    qoff = int(rng.integers(1, n - m + 1))
    planted = [int(p) for p in rng.integers(1, n - m + 1, 3)]
    q = datagen.plant_copies(s, qoff, m, planted, [1.0, float(rng.uniform(0.7, 1.4)), 1.0], [0.0, float(rng.uniform(-2, 2)), 0.1],
                             float(rng.uniform(0.001, 0.05)), seed=c)
    ctx.load_series(s)
    mean_len = float(rng.choice([2.0, 20.0, 64.0, 150.0, 700.0]))
    L, R = datagen.gen_intervals(n, m, int(rng.integers(300, 6000)), mean_len, must_cover=[qoff] + planted, seed=c)
    eps = float(rng.choice([0.02, 0.1, 0.3, 1.0])) * np.sqrt(m) + float(rng.uniform(0, 0.3))
    alpha = float(rng.choice([1.02, 1.2, 2.0, 50.0])); beta = float(rng.choice([0.05, 0.5, 5.0, 1e6]))
    o1 = oracle.rsm_ed(s, q, L, R, 0, eps); o2 = oracle.cnsm_ed(s, q, L, R, 0, eps, alpha, beta)
    for ab in (0, 1):
        ctx.set_option(kvmatch.OPT_EARLY_ABANDON, ab)
        for k in (3, 1):   # (3 the correlation-form pruner = the default, 1 the wave-per-candidate cross-check)
            ctx.set_option(kvmatch.OPT_ED_KERNEL, k)
            tag = f"case {c} n={n} m={m} ml={mean_len} eps={eps:.2f} a={alpha} b={beta} ab={ab} k={k}"
            same(kvmatch.QueryEngine(ctx).verify((L, R), 1, q, eps), o1, "rsm " + tag)
            same(kvmatch.NormQueryEngine(ctx).verify((L, R), 1, q, eps, alpha, beta), o2, "cnsm " + tag)
print("cases", cases, "verifications", cases * 8, "mismatches", bad, "answers(last)", len(o1[0]), len(o2[0]))
sys.exit(1 if bad else 0)

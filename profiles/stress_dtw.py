#!/usr/bin/env python3
"""One-off stress run (not part of the test suite): random DTW verifications (RSM-DTW and cNSM-DTW, cascade on and off,
every form of the DP: KVM_OPT_DP_KERNEL 0 .. 3) against the oracle, bit for bit.  `stress_dtw.py [cases]`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, ROOT)
import numpy as np, kvmatch
from kvmatch import datagen
from oracle import oracle
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(os.environ.get("SEED", "5")))
base = datagen.gen_series(2_000_000)
ctx = kvmatch.Phase2Context(0)
ctx.set_option(kvmatch.OPT_SORT_VIOLATION, 1); oracle.set_lenient_sort(True)  # compare the arithmetic also where the reference's sort throws
bad = 0
def same(a, o, tag):
    global bad
    oo, od, oc = o
    ok = len(a) == len(oo) and np.array_equal(a.offsets, oo) and np.array_equal(a.distances, od) and a.cnt_candidate == oc["cnt_candidate"]
    if not ok:
        bad += 1
        print("MISMATCH", tag, len(a), len(oo))
tot_answers = 0
for c in range(cases):
    n = int(rng.integers(30_000, 150_000))
    kind = c % 3
    if kind == 0: s = base[int(rng.integers(0, 1_500_000)):][:n].copy()
    elif kind == 1: s = np.cumsum(rng.uniform(-1, 1, n)) + rng.uniform(-3000, 3000)
    else: s = 3 * np.sin(np.arange(n) * rng.uniform(0.0005, 0.05)) + rng.normal(0, 0.2, n)
    m = int(rng.choice([64, 200, 513, 1024, 2000, 3000, 4096]))
    rho = int(rng.choice([2, 3, max(2, m // 40), max(2, m // 20), max(2, m // 10), max(2, m // 4)]))  # (<= 1 would mean a fraction of m)
    qoff = int(rng.integers(1, n - m + 1))
    planted = [int(p) for p in rng.integers(1, n - m + 1, 3)]
    q = datagen.plant_copies(s, qoff, m, planted, [1.0, float(rng.uniform(0.7, 1.4)), 1.0], [0.0, float(rng.uniform(-2, 2)), 0.1],
                             float(rng.uniform(0.001, 0.05)), seed=c)
    ctx.load_series(s)
    mean_len = float(rng.choice([2.0, 20.0, 64.0, 300.0]))
    L, R = datagen.gen_intervals(n, m, int(rng.integers(200, 3000)), mean_len, must_cover=[qoff] + planted, seed=c)
    eps = float(rng.choice([0.02, 0.1, 0.3])) * np.sqrt(m) + float(rng.uniform(0, 0.3))
    alpha = float(rng.choice([1.05, 1.5, 50.0])); beta = float(rng.choice([0.1, 2.0, 1e6]))
    o1 = oracle.rsm_dtw(s, q, L, R, 0, eps, rho); o2 = oracle.cnsm_dtw(s, q, L, R, 0, eps, rho, alpha, beta)
    tot_answers += len(o1[0]) + len(o2[0])
    for cascade in (1, 0):
        ctx.set_option(kvmatch.OPT_LB_CASCADE, cascade)
        for dpk in (0, 1, 2, 3):   # auto, LDS form, one candidate per wave, packed form
            ctx.set_option(kvmatch.OPT_DP_KERNEL, dpk)
            tag = f"case {c} n={n} m={m} rho={rho} ml={mean_len} eps={eps:.2f} a={alpha} b={beta} cascade={cascade} dp={dpk}"
            same(kvmatch.QueryEngineDtw(ctx).verify((L, R), 1, q, eps, rho), o1, "rsm " + tag)
            same(kvmatch.NormQueryEngineDtw(ctx).verify((L, R), 1, q, eps, rho, alpha, beta), o2, "cnsm " + tag)
    if c % 10 == 9: print("  ...", c + 1, "cases, mismatches", bad, flush=True)
print("cases", cases, "mismatches", bad, "oracle answers total", tot_answers)
sys.exit(1 if bad else 0)

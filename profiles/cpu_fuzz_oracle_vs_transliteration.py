#!/usr/bin/env python3
"""CPU only: the C oracle against the independent Python transliteration (tests/java_transliteration.py) on malformed
interval lists -- overlapping, nested, repeated, reversed, out-of-range, optionally unsorted -- on well-behaved and on
integer-valued series, all four engines, bit for bit incl. the IllegalArgument cases.  (The GPU fuzzers compare the library
with the oracle on the same kind of list; this one checks that the two readings of the Java text agree there too.)
`cpu_fuzz_oracle_vs_transliteration.py [cases] [seed]`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle
import java_transliteration as J
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
oracle.set_lenient_sort(False)
bad = total = exc = crash = 0
for case in range(cases):
    n = int(rng.choice([1500, 4000]))
    kind = case % 4
    s = (np.cumsum(rng.normal(size=n)) if kind == 0 else rng.integers(-3, 4, n).astype(np.float64) if kind == 1
         else np.round(np.cumsum(rng.normal(size=n))) * 0.5 if kind == 2 else 1e6 + np.cumsum(rng.normal(size=n)))
    m = int(rng.choice([25, 40, 64]))
    qoff = int(rng.integers(1, n - m)); q = s[qoff - 1:qoff - 1 + m].copy()
    if q.std() == 0: q[0] += 1
    k = int(rng.integers(0, 10))
    centres = rng.integers(-30, n + 30, k)
    if k and rng.random() < 0.7: centres[rng.integers(0, k)] = qoff
    L = centres - rng.integers(0, 40, k); R = centres + rng.integers(-5, 60, k)
    if rng.random() < 0.8:
        o = np.argsort(L, kind="stable"); L, R = L[o], R[o]
    L, R = L.astype(np.int32), R.astype(np.int32)
    ls = int(rng.choice([1, 1, 2, 3])); shift = (ls - 1) * 25
    eps = float(rng.choice([0.0, 0.5, 2.0, 3.0, 50.0])); rho = int(rng.choice([0, 2, m // 10]))
    a = float(rng.choice([1.0, 1.5, 1e9])); b = float(rng.choice([0.0, 2.0, 1e9]))
    sl, ql, Ll, Rl = s.tolist(), q.tolist(), [int(x) for x in L], [int(x) for x in R]
    for name, py, c in (
            ("rsm_ed", lambda: J.rsm_ed(sl, ql, Ll, Rl, ls, eps), lambda: oracle.rsm_ed(s, q, L, R, shift, eps)),
            ("cnsm_ed", lambda: J.cnsm_ed(sl, ql, Ll, Rl, ls, eps, a, b), lambda: oracle.cnsm_ed(s, q, L, R, shift, eps, a, b)),
            ("rsm_dtw", lambda: J.rsm_dtw(sl, ql, Ll, Rl, ls, eps, rho), lambda: oracle.rsm_dtw(s, q, L, R, shift, eps, rho)),
            ("cnsm_dtw", lambda: J.cnsm_dtw(sl, ql, Ll, Rl, ls, eps, rho, a, b), lambda: oracle.cnsm_dtw(s, q, L, R, shift, eps, rho, a, b))):
        total += 1
        try:
            ans, cnt = py(); pres = ([x[0] for x in ans], [x[1] for x in ans], cnt)
        except J.IllegalArgumentException:
            pres = "EXC"
        except J.ArrayIndexOutOfBoundsException:
            # DtwUtils.lowerUpperLemire on fetched data no longer than the band (an interval beyond the last window start, or
            # rho >= the fetched length): the reference ends in ArrayIndexOutOfBoundsException; the oracle and the library
            # go on (not replicated).  Nothing to compare.
            crash += 1
            continue
        try:
            o = c(); cres = (o[0].tolist(), o[1].tolist(), o[2]["cnt_candidate"])
        except oracle.OracleError:
            cres = "EXC"
        if pres == "EXC" or cres == "EXC": exc += 1
        if pres != cres:
            bad += 1
            print(f"DIFFER case {case} {name} kind={kind} m={m} ls={ls} eps={eps} rho={rho} a={a} b={b} k={k}: {str(pres)[:90]} | {str(cres)[:90]}", flush=True)
print("engine calls", total, "with an exception on either side", exc, "where the reference ends in ArrayIndexOutOfBounds (skipped)", crash, "differences", bad)
sys.exit(1 if bad else 0)
